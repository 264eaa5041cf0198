#!/usr/bin/env python
"""Diagnostic builds of the library with parts of syrk_flat_kernel compiled out (results WRONG by construction): what
update() owes to the flat kernel's LDS operand reads, LDS-DMA pieces and MFMAs - the question behind "would a larger
per-wave tile (fewer operand bytes per MFMA) pay".  Builds tools/micro/libcurv_flat_ab<mask>.so for every mask given
(default 0 1 2 4 8; mask bits in csrc/syrk_flat.hip); the other objects come from csrc/build/ (run build() first).
    python tools/make_flat_ablate.py && gpurun -- 'bash tools/ab_libs.sh 2 tools/micro/libcurv_flat_ab0.so tools/micro/libcurv_flat_ab1.so ...'"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvature_amd", "csrc")
masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8]
objs = [o for o in glob.glob(os.path.join(CSRC, "build", "*.o")) if not o.endswith("syrk_flat.hip.o")]
assert objs, "run __graft_entry__.build() first"
for m in masks:
    obj = os.path.join(ROOT, "tools", "micro", f"syrk_flat_ab{m}.o")
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                           f"-DCURV_FLAT_ABLATE={m}", "-c", "-o", obj, os.path.join(CSRC, "syrk_flat.hip")])
    out = os.path.join(ROOT, "tools", "micro", f"libcurv_flat_ab{m}.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out, obj] + objs + ["-ldl"])
    print("built", out)
