#!/usr/bin/env python
"""Per-workgroup timeline of the far outer_update_kernel launch of one panel (diagnostics only).
`build` writes tools/micro/libcurv_oprof.so (invert.hip + wall-clock probes); `run [k0]` inverts three
4608^2 factors and prints where a workgroup's life goes and how many run side by side."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "micro", "libcurv_oprof.so")
NREC = 16384


def sub(s, a, b):
    assert a in s, a
    return s.replace(a, b, 1)


def build():
    s = open(os.path.join(CSRC, "invert.hip")).read()
    s = sub(s, "constexpr int OKS = 16;", "__device__ unsigned long long g_ot[8 * %d];\n__device__ int g_probe_k0 = 32;\n"
            "__device__ unsigned long long g_t_first, g_t_loop;   // scratch of the probing workgroup's thread 0\n"
            "constexpr int OKS = 16;" % NREC)
    # inside the tile core: time of the first barrier (first operands in LDS) and of the end of the K loop,
    # kept in registers of thread 0 and handed back through two by-reference arguments
    s = sub(s, "__device__ __forceinline__ void tile_product_k32(const TileJob& o, double* __restrict__ As, double* __restrict__ Bs) {",
            "__device__ __forceinline__ void tile_product_k32(const TileJob& o, double* __restrict__ As, double* __restrict__ Bs,\n"
            "                                                 unsigned long long* t_first = nullptr, unsigned long long* t_loop = nullptr) {")
    s = sub(s, "    __syncthreads();\n    if (ke + OKS < ke1) fetch(ke + OKS);", "    __syncthreads();\n    if (t_first && ke == ke0) *t_first = wall_clock64();\n    if (ke + OKS < ke1) fetch(ke + OKS);")
    s = sub(s, "  if constexpr (WV == 2) {\n    store_acc(o.C, np, acc, wm, wn, lane, o.mode);", "  if (t_loop) *t_loop = wall_clock64();\n  if constexpr (WV == 2) {\n    store_acc(o.C, np, acc, wm, wn, lane, o.mode);")
    s = sub(s, "                                                  int strip, int n_items, double* __restrict__ As, double* __restrict__ Bs) {\n",
            "                                                  int strip, int n_items, double* __restrict__ As, double* __restrict__ Bs) {\n"
            "  const unsigned long long w0 = wall_clock64();\n  unsigned long long w1 = 0, w2 = 0;\n")
    s = sub(s, "  tile_product_k32<WV>(o, As, Bs);\n}\n__global__ void __launch_bounds__(INV_THREADS, 3)\nouter_update_kernel(",
            "  tile_product_k32<WV>(o, As, Bs, &w1, &w2);\n"
            "  if (!strip && k0 == g_probe_k0 && threadIdx.x == 0 && blockIdx.x < %d) {\n"
            "    unsigned long long* q = g_ot + 8 * blockIdx.x;\n"
            "    q[0] = w0; q[1] = w1; q[2] = w2; q[3] = wall_clock64();\n"
            "    q[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);\n"
            "    q[6] = ((unsigned long long)(trailing ? 1 : 0) << 40) | ((unsigned long long)i << 20) | (unsigned)j; q[7] = (o.ke1 - o.ke0) / OKS;\n"
            "  }\n}\n__global__ void __launch_bounds__(INV_THREADS, 3)\nouter_update_kernel(" % NREC)
    s += '''
extern "C" int curv_debug_outer_times(unsigned long long* out, int k0) {
  if (out == nullptr) return (int)hipMemcpyToSymbol(HIP_SYMBOL(curv::g_probe_k0), &k0, sizeof(int));
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(curv::g_ot), 8 * %d * sizeof(unsigned long long));
}
''' % NREC
    src = "/tmp/invert_oprof.hip"
    open(src, "w").write(s)
    others = ["api.cpp", "elementwise.hip", "syrk.hip", "gemm.hip", "inf.hip", "eigh.hip"]
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-o", OUT, src] + [os.path.join(CSRC, o) for o in others]
    subprocess.check_call(cmd)
    print("built", OUT)


def run(k0, nfac):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from curvature_amd import _lib
    _lib.LIB_PATH = OUT
    _lib._stale = lambda: False
    from curvature_amd import ops
    h = _lib.lib()
    h.curv_debug_outer_times.restype = ctypes.c_int
    h.curv_debug_outer_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    Fs = []
    for i in range(nfac):
        torch.manual_seed(i)
        X = torch.randn(4608, 4096, device=dev)
        Fs.append((X @ X.t() / 4096).contiguous())
    h.curv_debug_outer_times(None, k0)
    for _ in range(3):
        ops.chol_inv_lower(Fs, [1.0] * nfac, [1000.0] * nfac, check=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * NREC))()
    h.curv_debug_outer_times(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NREC, 8).astype(np.int64)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    tick = 0.01   # us per wall-clock tick (100 MHz)
    st, first, loop_end, end = [(a[:, c] - t0) * tick for c in range(4)]
    print(f"k0={k0}: {len(a)} workgroups recorded, launch span {end.max():.1f} us")
    print(f"  life {np.mean(end - st):.2f} us (p10 {np.percentile(end - st, 10):.2f}, p90 {np.percentile(end - st, 90):.2f}); "
          f"to first data {np.mean(first - st):.2f}; K loop {np.mean(loop_end - first):.2f} ({np.mean((loop_end - first) / np.maximum(a[:, 7], 1)):.2f} per step); "
          f"read-modify-write {np.mean(end - loop_end):.2f}")
    hw, xcc = a[:, 4], a[:, 5] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
    key = xcc * 1024 + cu
    print(f"  distinct (xcc, se, sh, cu): {len(np.unique(key))}; workgroups per XCC: {np.bincount(xcc, minlength=8).tolist()}")
    # concurrency: average number of live workgroups over the launch span, overall and per CU
    span = end.max() - st.min()
    print(f"  mean live workgroups {np.sum(end - st) / span:.0f} = {np.sum(end - st) / span / max(len(np.unique(key)), 1):.2f} per CU")
    # first-start skew and tail
    order = np.argsort(st)
    print(f"  starts: 10% by {np.percentile(st, 10):.1f} us, 50% by {np.percentile(st, 50):.1f}, 90% by {np.percentile(st, 90):.1f}, last {st.max():.1f}; "
          f"ends: 50% by {np.percentile(end, 50):.1f}, 90% by {np.percentile(end, 90):.1f}, 99% by {np.percentile(end, 99):.1f}")
    for lo in range(0, int(span) + 1, max(int(span) // 10, 1)):
        live = np.sum((st <= lo) & (end > lo))
        print(f"    t={lo:4d} us live {live}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
