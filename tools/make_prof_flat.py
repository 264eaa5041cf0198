#!/usr/bin/env python
"""Build tools/micro/libcurv_prof_flat.so: the library with clock64() probes patched into flat_body of syrk_flat.hip
(wait at the top of a stage = vmcnt(0) + barrier, vs the rest of the stage).  Diagnostics only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")


def sub(s, a, b):
    assert a in s, a
    return s.replace(a, b, 1)


def main():
    s = open(os.path.join(CSRC, "syrk_flat.hip")).read()
    s = sub(s, "namespace curv {\n", "namespace curv {\n__device__ unsigned long long g_fprof[16];\n__device__ unsigned long long g_ftimes[3 * 32768];\n")
    s = sub(s, "  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};\n",
            "  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};\n  long long tprev = clock64(); const unsigned long long wall0 = wall_clock64();\n"
            "  unsigned long long a_wait = 0, a_bar = 0, a_work = 0, a_pro = 0; unsigned nst = 0;\n")
    s = sub(s, "  for (int t = t0; t < t1; ++t) {\n    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's DMA of stage t has landed\n    __syncthreads();",
            "  { long long t_ = clock64(); a_pro += t_ - tprev; tprev = t_; }\n"
            "  for (int t = t0; t < t1; ++t) {\n    { long long t_ = clock64(); a_work += t_ - tprev; tprev = t_; }\n"
            "    __builtin_amdgcn_s_waitcnt(0x0f70);\n    { long long t_ = clock64(); a_wait += t_ - tprev; tprev = t_; }\n"
            "    __syncthreads();\n    { long long t_ = clock64(); a_bar += t_ - tprev; tprev = t_; ++nst; }")
    s = sub(s, "  gfloat_t* slab = (gfloat_t*)slabs + d.slab_base + (long long)local * (TM * TM);\n",
            "  { long long t_ = clock64(); a_work += t_ - tprev; tprev = t_; }\n"
            "  if (lane == 0) { atomicAdd(&g_fprof[0], a_pro); atomicAdd(&g_fprof[1], a_wait); atomicAdd(&g_fprof[2], a_bar); atomicAdd(&g_fprof[3], a_work);"
            " atomicAdd(&g_fprof[4], (unsigned long long)nst); atomicAdd(&g_fprof[5], 1ull); }\n"
            "  if (tid == 0 && blockIdx.x < 32768) { g_ftimes[3 * blockIdx.x] = wall0; g_ftimes[3 * blockIdx.x + 1] = wall_clock64();"
            " g_ftimes[3 * blockIdx.x + 2] = ((unsigned long long)d.dim << 32) | (unsigned)(d.W); }\n"
            "  gfloat_t* slab = (gfloat_t*)slabs + d.slab_base + (long long)local * (TM * TM);\n")
    s += '''
extern "C" int curv_debug_flat_times(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(curv::g_ftimes), 3 * 32768 * sizeof(unsigned long long));
}
extern "C" int curv_debug_flat_prof(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(curv::g_fprof), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(curv::g_fprof), 16 * sizeof(unsigned long long));
}
'''
    src = "/tmp/syrk_flat_prof.hip"
    open(src, "w").write(s)
    out = os.path.join(ROOT, "tools", "micro", "libcurv_prof_flat.so")
    others = ["api.cpp", "elementwise.hip", "syrk.hip", "syrk_corr.hip", "syrk_pre.hip", "syrk_small.hip", "collective.cpp", "invert.hip", "gemm.hip", "inf.hip", "eigh.hip"]
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-o", out, src] + [os.path.join(CSRC, o) for o in others]
    subprocess.check_call(cmd)
    print("built", out)


if __name__ == "__main__":
    sys.exit(main())
