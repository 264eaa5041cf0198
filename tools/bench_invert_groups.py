#!/usr/bin/env python
"""Invert time of the ResNet-50 factors split the way chol_sweep splits them: big group alone, small group
alone, everything (tells whether the batched sweep is chain-bound or throughput-bound)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, ops  # noqa: E402


def run(tag, sizes, iters=5):
    dev = torch.device("cuda:0")
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
    flops = sum((2.0 / 3.0) * n ** 3 for n in sizes)
    dt = (time.perf_counter() - t0) / iters
    print(f"{tag}: {len(sizes)} factors {dt * 1e3:.3f} ms  {flops / dt / 1e12:.1f} TFLOP/s algorithmic")


if __name__ == "__main__":
    rows = models.layer_table(models.resnet50(), (3, 224, 224))
    sizes = []
    for r in rows:
        sizes += [r["n"], r["m"]]
    big = [n for n in sizes if n > 2304]
    small = [n for n in sizes if n <= 2304]
    run("big", big)
    run("small", small)
    run("mid(2304,2048)", [n for n in sizes if 2048 <= n <= 2304])
    run("<=1152", [n for n in sizes if n <= 1152])
    run("all", sizes)
