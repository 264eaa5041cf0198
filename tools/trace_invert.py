#!/usr/bin/env python
"""Invert-only loop on the ResNet-50 factor sizes (for rocprofv3 --kernel-trace timelines)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import _lib  # noqa: E402

if os.environ.get("CURV_ALT_LIB"):                       # A/B of an alternative build on one box
    _lib.LIB_PATH = os.path.abspath(os.environ["CURV_ALT_LIB"])
from curvature_amd import models, ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda:0")
    rows = models.layer_table(models.resnet50(), (3, 224, 224))
    sizes = []
    for r in rows:
        sizes += [r["n"], r["m"]]
    Fs = []
    for i, n in enumerate(sizes):
        torch.manual_seed(i)
        k = min(n + 8, 4096)
        X = torch.randn(n, k, device=dev)
        Fs.append((X @ X.t() / k).contiguous())
    add, mul = [1.0] * len(Fs), [1000.0] * len(Fs)
    for _ in range(2):
        ops.chol_inv_lower(Fs, add, mul, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.chol_inv_lower(Fs, add, mul, check=False)
    torch.cuda.synchronize()
    print(f"invert: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms over {len(Fs)} factors")


if __name__ == "__main__":
    main()
