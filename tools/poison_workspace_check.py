#!/usr/bin/env python
"""Does the factor build read workspace memory it has not written?  Every geometry of tests/test_syrk_gpu.py with the
library's scratch buffers filled with NaN bit patterns in front of every call (first build and accumulation), results
checked for finiteness and against a clean run bit for bit.  A flaky non-finite factor in the test suite is what this hunts."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from curvature_amd import _lib  # noqa: E402
if os.environ.get("CURV_ALT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CURV_ALT_LIB"])
from curvature_amd import ops  # noqa: E402
from test_syrk_gpu import CONV_CASES  # noqa: E402

dev = torch.device("cuda:0")
bad = 0
for idx, case in enumerate(CONV_CASES):
    N, C, H, W, k, s, p, bias = case
    k2 = (k, k) if isinstance(k, int) else k
    s2 = (s, s) if isinstance(s, int) else s
    p2 = (p, p) if isinstance(p, int) else p
    torch.manual_seed(1234)
    x = torch.relu(torch.randn(N, C, H, W)).to(dev)
    Ho = (H + 2 * p2[0] - k2[0]) // s2[0] + 1
    Wo = (W + 2 * p2[1] - k2[1]) // s2[1] + 1
    g = (torch.randn(N, 37, Ho, Wo) / N).to(dev)
    n = C * k2[0] * k2[1] + int(bias)
    results = []
    for poison in (False, True, True):
        A = torch.full((n, n), float("nan"), device=dev)
        G = torch.full((37, 37), float("nan"), device=dev)
        jobs = [ops.FactorJob(x, A, k2, s2, p2, bias, 1.0 / (N * Ho * Wo), True),
                ops.FactorJob(g, G, (1, 1), (1, 1), (0, 0), False, N / (Ho * Wo), True)]
        for rnd in range(2):
            if poison:
                torch.cuda.synchronize()
                for buf in list(ops._workspaces.values()):
                    buf.fill_(0xFF)
                ops._kfac_last.clear()        # (the head of the workspace is gone too: no resident descriptor table)
            ops.kfac_accumulate(jobs)
            for j in jobs:
                j.first = False
        torch.cuda.synchronize()
        results.append((A.clone(), G.clone()))
    ok = all(bool(torch.isfinite(t).all()) for r in results for t in r)
    same = all(torch.equal(results[0][i], results[k][i]) for k in (1, 2) for i in (0, 1))
    if not (ok and same):
        bad += 1
    print(f"case {idx:2d} {case}: finite {ok}, poisoned == clean {same}", flush=True)
print("FAILED" if bad else "all clean", bad)
