#!/usr/bin/env python
"""Ablation builds of outer_update_kernel (diagnostics): `build` writes tools/micro/libcurv_abl{0,1,2,3}.so
(0 = as shipped, 1 = no MFMA, 2 = no global loads in the K loop, 3 = neither; the patch goes into the shared
tile core, so the panel products are ablated too), `run <v>` inverts three
4608^2 factors with variant v (to be timed under rocprofv3 --kernel-trace, see tools/ablate_outer.sh)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")


def sub(s, a, b):
    assert a in s, a
    return s.replace(a, b, 1)


def build():
    base = open(os.path.join(CSRC, "invert.hip")).read()
    for v in range(4):
        s = base
        if v & 1:
            s = sub(s, "        for (int n = 0; n < T; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);\n",
                    "        for (int n = 0; n < T; ++n) acc[m][n][0] += a[m] * b[n];\n")
        if v & 2:
            s = sub(s, "    if (ke + OKS < ke1) fetch(ke + OKS);\n", "")
        src = f"/tmp/invert_abl{v}.hip"
        open(src, "w").write(s)
        out = os.path.join(ROOT, "tools", "micro", f"libcurv_abl{v}.so")
        others = ["api.cpp", "elementwise.hip", "syrk.hip", "gemm.hip", "inf.hip", "eigh.hip"]
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-I" + CSRC, "-o", out, src] + [os.path.join(CSRC, o) for o in others]
        subprocess.check_call(cmd)
        print("built", out)


def run(v):
    sys.path.insert(0, ROOT)
    import torch
    from curvature_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "micro", f"libcurv_abl{v}.so")
    _lib._stale = lambda: False
    from curvature_amd import ops
    dev = torch.device("cuda:0")
    Fs = []
    for i in range(3):
        torch.manual_seed(i)
        X = torch.randn(4608, 4096, device=dev)
        Fs.append((X @ X.t() / 4096).contiguous())
    for _ in range(3):
        ops.chol_inv_lower(Fs, [1.0] * 3, [1000.0] * 3, check=False)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]))
