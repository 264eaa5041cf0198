#!/usr/bin/env python
"""Is invert() of the ResNet-50 factors paced by the host?  The calls are enqueued behind a long spin kernel, so that the host
has finished enqueueing all of them before the GPU starts the first; GPU time per call from events."""
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
    for spin_ms in (0, 60):
        calls = 4
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(calls + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if spin_ms:
            torch.cuda._sleep(int(spin_ms * 2.1e6))          # ~2.1 GHz cycles
        ev[0].record()
        for c in range(calls):
            ops.chol_inv_lower(Fs, add, mul, check=False)
            ev[c + 1].record()
        t_host = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        per = [ev[c].elapsed_time(ev[c + 1]) for c in range(calls)]
        print(f"spin {spin_ms} ms in front: host enqueued {calls} calls in {t_host:.2f} ms; GPU ms per call: " + " ".join(f"{p:.2f}" for p in per))


if __name__ == "__main__":
    main()
