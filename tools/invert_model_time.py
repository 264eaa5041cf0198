#!/usr/bin/env python
"""invert() of the 108 ResNet-50 factor sizes (synthetic well-conditioned factors) and of a few rank shares, median of
back-to-back checked calls; for A/B of library builds (CURV_ALT_LIB)."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvature_amd import _lib  # noqa: E402

if os.environ.get("CURV_ALT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CURV_ALT_LIB"])
import torch  # noqa: E402
from curvature_amd import models, ops  # noqa: E402

dev = torch.device("cuda:0")
_lib.init_streams(dev)


def factors(sizes):
    Fs = []
    for i, n in enumerate(sizes):
        torch.manual_seed(i)
        k = min(n + 8, 4096)
        X = torch.randn(n, k, device=dev)
        Fs.append((X @ X.t() / k).contiguous())
    return Fs


def run(name, sizes, iters=15):
    Fs = factors(sizes)
    add, mul = [1.0] * len(Fs), [1000.0] * len(Fs)
    for _ in range(3):
        ops.chol_inv_lower(Fs, add, mul)
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.chol_inv_lower(Fs, add, mul)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: {statistics.median(ts):.3f} ms (min {min(ts):.3f})", flush=True)


rows = models.layer_table(models.resnet50(), (3, 224, 224))
sizes = []
for r in rows:
    sizes += [r["n"], r["m"]]
which = sys.argv[1:] or ["model", "4608", "3x4608", "2304"]
if "model" in which:
    run("resnet50 108 factors", sizes)
if "4608" in which:
    run("one 4608", [4608, 512])
if "3x4608" in which:
    run("three 4608", [4608, 4608, 4608])
if "2304" in which:
    run("2304", [2304, 256])
