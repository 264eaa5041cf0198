#!/usr/bin/env python
"""invert() of the 42 ResNet-18 factor sizes (synthetic well-conditioned factors), median of checked calls - which launch form
of the sweep suits a mid-sized model (CURV_LATENCY_MAX: calls with at most this many factors take the chain-bound forms)."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvature_amd import _lib, models, ops  # noqa: E402

dev = torch.device("cuda:0")
_lib.init_streams(dev)
rows = models.layer_table(getattr(models, sys.argv[1] if len(sys.argv) > 1 else "resnet18")(), (3, 224, 224))
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
    ops.chol_inv_lower(Fs, add, mul)
ts = []
for _ in range(15):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.chol_inv_lower(Fs, add, mul)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"{len(Fs)} factors (largest {max(sizes)}): {statistics.median(ts):.3f} ms (min {min(ts):.3f})")
