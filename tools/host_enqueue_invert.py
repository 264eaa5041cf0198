#!/usr/bin/env python
"""Host time to ENQUEUE one whole-model inversion (no synchronisation) against its GPU time: is invert() launch-bound?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rows = models.layer_table(models.resnet50(), (3, 224, 224))
    sizes = []
    for r in rows:
        sizes += [r["n"], r["m"]]
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "big":
        sizes = [n for n in sizes if n > 2304]
    elif which == "small":
        sizes = [n for n in sizes if n <= 2304]
    elif which != "all":
        sizes = [int(v) for v in which.split(",")]
    Fs = []
    for i, n in enumerate(sizes):
        torch.manual_seed(i)
        k = min(n + 8, 4096)
        X = torch.randn(n, k, device=dev)
        Fs.append((X @ X.t() / k).contiguous())
    add, mul = [1.0] * len(Fs), [1000.0] * len(Fs)
    for _ in range(3):
        ops.chol_inv_lower(Fs, add, mul, check=False)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.chol_inv_lower(Fs, add, mul, check=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    print(f"{which}: {len(Fs)} factors: host enqueue {min(host):.3f} ms, call + sync {min(total):.3f} ms")


if __name__ == "__main__":
    main()
