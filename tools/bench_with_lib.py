#!/usr/bin/env python
"""bench.py against an alternative build of the library (A/B of kernel variants on one box):
    CURV_ALT_LIB=tools/micro/libcurv_x.so python tools/bench_with_lib.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvature_amd import _lib  # noqa: E402

if os.environ.get("CURV_ALT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CURV_ALT_LIB"])
import bench  # noqa: E402

bench.main()
