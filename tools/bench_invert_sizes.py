#!/usr/bin/env python
"""Latency of the batched invert for a few factor-size lists (what one rank owns under layer sharding)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import ops  # noqa: E402


def run(sizes, iters=5):
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
    print(f"{sizes}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms")


if __name__ == "__main__":
    # CURV_DUMMY_STREAMS=k: k unused streams created before the sweep's own (what RCCL / a data loader would add; the
    # per-step launches of the sweep were sensitive to that: profiles/r03_stream_sensitivity.txt)
    dummies = [torch.cuda.Stream() for _ in range(int(os.environ.get("CURV_DUMMY_STREAMS", "0")))]
    for sizes in ([4608], [4608, 512], [2304], [2304, 256], [1024], [4608, 4608, 4608], [2048, 512, 1024, 256]):
        run(sizes)
