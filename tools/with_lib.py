#!/usr/bin/env python
"""Run any script of this tree against an alternative build of the library (same-box A/B of kernel variants):
    python tools/with_lib.py tools/micro/libcurv_x.so tools/bench_syrk.py --iters 20"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvature_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
