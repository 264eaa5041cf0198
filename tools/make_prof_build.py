#!/usr/bin/env python
"""Build tools/micro/libcurv_prof.so: the library with clock64() probes patched into syrk_body (phase
breakdown for tools/prof_syrk.py).  Diagnostics only; the patched source lives in /tmp."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")


def sub(s, a, b):
    assert a in s, a
    return s.replace(a, b, 1)


def main():
    s = open(os.path.join(CSRC, "syrk.hip")).read()
    s = sub(s, "namespace curv {\n", "namespace curv {\n__device__ unsigned long long g_prof[16];\n__device__ unsigned long long g_times[3 * 16384];\n"
            "#define PROF(k) { long long t_ = clock64(); pacc[k] += (unsigned)(t_ - tprev); tprev = t_; }\n")
    s = sub(s, "  const int h = lane >> 5;\n", "  const int h = lane >> 5;\n  long long tprev = clock64();\n  const unsigned long long wall0 = wall_clock64();\n"
            "  unsigned pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned nmf = 0, pacc8 = 0, nchunk = 0;\n")
    s = sub(s, "  for (int ch = ch_begin; ch < ch_end; ++ch) {\n", "  PROF(0)\n  for (int ch = ch_begin; ch < ch_end; ++ch) {\n")
    s = sub(s, "#ifdef CURV_DIAG\n    if (!(d.pad0 & 10) || ch == ch_begin) store_stage(cur);\n#else\n    store_stage(cur);\n#endif\n    __syncthreads();\n",
            "    PROF(1)\n    __builtin_amdgcn_s_waitcnt(0x0f70);\n    { long long t_ = clock64(); pacc8 += (unsigned)(t_ - tprev); tprev = t_; }\n"
            "    if (!(d.pad0 & 10) || ch == ch_begin) store_stage(cur);\n    PROF(2)\n    __syncthreads();\n    PROF(3)\n    ++nchunk;\n")
    s = sub(s, "    // ---- MFMA over this wave's share of the chunk ----", "    PROF(4)")
    s = sub(s, "      else mfma_chunk(std::integral_constant<int, 0>{}, work, by_sample);     // strides > 2: step offsets at run time\n    }\n    __syncthreads();\n",
            "      else mfma_chunk(std::integral_constant<int, 0>{}, work, by_sample);\n"
            "      nmf += work.ns * work.ra * ((work.wa + 1) >> 1) / KSTRIDE * (part == 0 ? 4 : part == 1 ? 3 : 2);\n    }\n"
            "    PROF(5)\n    __syncthreads();\n    PROF(6)\n")
    s = sub(s, "      if (part != 2) q[(32 + row) * 128 + 32 + r32] = acc11[reg];\n    }\n  }\n}\n",
            "      if (part != 2) q[(32 + row) * 128 + 32 + r32] = acc11[reg];\n    }\n  }\n  PROF(7)\n"
            "  if (PRE && lane == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&g_prof[k], (unsigned long long)pacc[k]);"
            " atomicAdd(&g_prof[10], (unsigned long long)nmf); atomicAdd(&g_prof[11], 1ull);"
            " atomicAdd(&g_prof[8], (unsigned long long)pacc8); atomicAdd(&g_prof[12], (unsigned long long)nchunk); }\n"
            "  if (PRE && tid == 0 && blockIdx.x < 16384) { g_times[3 * blockIdx.x] = wall0; g_times[3 * blockIdx.x + 1] = wall_clock64();"
            " g_times[3 * blockIdx.x + 2] = ((unsigned long long)d.dim << 32) | (unsigned)(d.kh * 100 + d.TM); }\n}\n")
    s += '''
extern "C" int curv_debug_syrk_times(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(curv::g_times), 3 * 16384 * sizeof(unsigned long long));
}
extern "C" int curv_debug_syrk_prof(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(curv::g_prof), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(curv::g_prof), 16 * sizeof(unsigned long long));
}
'''
    if os.environ.get("PROF_ONE_WG"):
        # one workgroup per CU: pad the static LDS past half of the 160 KB
        s = sub(s, "constexpr int PATCH_WORDS = 2 * PANEL_WORDS;", "constexpr int PATCH_WORDS = 2 * PANEL_WORDS + 4096;")
        s = sub(s, "static_assert(2 * SMEM_WORDS * 4 <= 160 * 1024", "static_assert(SMEM_WORDS * 4 <= 160 * 1024")
    src = "/tmp/syrk_prof.hip"
    open(src, "w").write(s)
    out = os.path.join(ROOT, "tools", "micro", "libcurv_prof.so")
    others = ["api.cpp", "elementwise.hip", "syrk_flat.hip", "syrk_corr.hip", "syrk_pre.hip", "syrk_small.hip", "collective.cpp", "invert.hip", "gemm.hip", "inf.hip", "eigh.hip"]
    # -DCURV_DIAG: the only build in which CURV_SYRK_ABLATE (phase ablation switches) is read
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DCURV_DIAG", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-o", out, src] + [os.path.join(CSRC, o) for o in others]
    subprocess.check_call(cmd)
    print("built", out)


if __name__ == "__main__":
    sys.exit(main())
