#!/usr/bin/env python
"""One source of the library recompiled with extra -D flags and linked with the product's other objects (csrc/build/, run
build() first) into tools/micro/libcurv_<tag>.so - for same-box A/B runs (tools/ab_libs.sh, tools/ab_update.py):
    python tools/make_variant.py syrk_flat.hip kc32 -DCURV_FLAT_KC=32"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")
src, tag, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
objs = [o for o in glob.glob(os.path.join(CSRC, "build", "*.o")) if os.path.basename(o) != src + ".o"]
assert len(objs) == len(glob.glob(os.path.join(CSRC, "build", "*.o"))) - 1, "run __graft_entry__.build() first"
obj = os.path.join(ROOT, "tools", "micro", f"{src}.{tag}.o")
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
                      + flags + ["-c", "-o", obj, os.path.join(CSRC, src)])
out = os.path.join(ROOT, "tools", "micro", f"libcurv_{tag}.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, obj] + objs + ["-ldl"])
print("built", out)
