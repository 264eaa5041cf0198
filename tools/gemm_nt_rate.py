#!/usr/bin/env python
"""Rate of gemm_nt_kernel (csrc/gemm.hip: both operands K-contiguous, LDS-DMA staging) on uniform problems: tells the
kernel's own efficiency from the shape effects of the samplers' launches (triangular K ranges, few long tiles).
Diagnostics only."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import ops  # noqa: E402


def run(M, N, K, reps=10):
    dev = torch.device("cuda:0")
    A = torch.randn(M, K, device=dev)
    Bt = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    plan = ops.GemmPlan([ops.Gemm(A, Bt.t(), C)])
    plan.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tiles = -(-M // 128) * -(-N // 128)
    print(f"M={M:5d} N={N:5d} K={K:5d}: {tiles:5d} tiles  {dt * 1e3:7.3f} ms  {2.0 * M * N * K / dt / 1e12:6.1f} TFLOP/s")


if __name__ == "__main__":
    for shape in ((2048, 4096, 2048), (4096, 4096, 4096), (2048, 4096, 512), (2048, 4096, 128), (8192, 8192, 1024),
                  (512, 4608, 4608), (1024, 4096, 2048)):
        run(*shape)
