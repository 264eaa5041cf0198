#!/usr/bin/env python
"""Does specialising syrk_patch_kernel per staging path pay?  `build` writes tools/micro/libcurv_lin.so (the
kernel with only the linear path compiled in) and libcurv_vec4.so (only the float4 flat path); `run <lib> <class>`
times one factor class with it (compare with the shipped library).  Diagnostics only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")


def sub(s, a, b):
    assert a in s, a
    return s.replace(a, b, 1)


def build():
    base = open(os.path.join(CSRC, "syrk.hip")).read()
    for name in ("lin", "vec4"):
        s = base
        if name == "lin":
            s = sub(s, "  const int compact = d.compact, vec4 = d.vec4;\n  const bool flat1 = d.flat && !vec4;",
                    "  const int compact = d.compact, vec4 = 0;\n  const bool flat1 = false;")
            s = sub(s, "  const int lin = d.lin;                               // 0, or V\n",
                    "  const int lin = d.lin;                               // 0, or V\n  __builtin_assume(lin != 0);\n")
        else:
            s = sub(s, "  const int compact = d.compact, vec4 = d.vec4;\n  const bool flat1 = d.flat && !vec4;",
                    "  const int compact = d.compact, vec4 = 1;\n  const bool flat1 = false;")
        src = f"/tmp/syrk_{name}.hip"
        open(src, "w").write(s)
        out = os.path.join(ROOT, "tools", "micro", f"libcurv_{name}.so")
        others = ["api.cpp", "elementwise.hip", "invert.hip", "gemm.hip", "inf.hip", "eigh.hip"]
        cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", 
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out, src] + [os.path.join(CSRC, o) for o in others]
        subprocess.check_call(cmd, cwd="/tmp")
        print("built", out)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        sys.path.insert(0, ROOT)
        from curvature_amd import _lib
        if sys.argv[2] != "base":
            _lib.LIB_PATH = os.path.join(ROOT, "tools", "micro", f"libcurv_{sys.argv[2]}.so")
            _lib._stale = lambda: False
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sys.argv = ["bench_syrk.py", "--only", sys.argv[3], "--iters", "10"]
        import bench_syrk
        bench_syrk.main()
