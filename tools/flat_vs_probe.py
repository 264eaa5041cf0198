#!/usr/bin/env python
"""The product's LDS-DMA factor-build kernel on the clean single-factor workloads of tools/micro/flat_shape_probe.hip
(one flattened factor, inputs streamed from HBM, ~1.5 s of back-to-back calls before the timed ones): executed
n (n + 1) K flops per call against the 157.3 TFLOP/s fp32 MFMA peak, next to what the probe's bare kernel reaches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import ops  # noqa: E402

PEAK = 157.3e12


def run(N, C, H, W, reps=20):
    dev = torch.device("cuda:0")
    torch.manual_seed(C + H)
    g = torch.randn(N, C, H, W, device=dev) - 0.3
    G = torch.zeros(C, C, device=dev)
    job = [ops.FactorJob(g, G, (1, 1), (1, 1), (0, 0), False, 1.0 / (N * H * W), True)]
    ops.kfac_accumulate(job)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    warm = 0
    while time.perf_counter() - t0 < 1.5:
        for _ in range(20):
            ops.kfac_accumulate(job)
        torch.cuda.synchronize()
        warm += 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.kfac_accumulate(job)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = C * (C + 1.0) * N * H * W
    print(f"C={C:5d} HW={H * W:5d} ({H}x{W}) N={N:4d}: {ms:7.3f} ms/call  {flops / ms / 1e9:6.1f} TFLOP/s executed "
          f"({flops / ms / 1e9 / 157.3:.3f} of 157.3)  [{warm} warm calls, input {g.numel() * 4 / 1e6:.0f} MB]", flush=True)


if __name__ == "__main__":
    cases = [(128, 1024, 28, 28), (512, 2048, 12, 16), (512, 2048, 14, 14), (1024, 2048, 7, 7), (128, 512, 56, 56),
             (512, 1024, 14, 14), (128, 256, 56, 56), (512, 4096, 8, 8)]
    only = [int(a) for a in sys.argv[1:]]
    for i, c in enumerate(cases):
        if not only or i in only:
            run(*c)
