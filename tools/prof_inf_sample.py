#!/usr/bin/env python
"""INF.sample_and_replace at ResNet-50 size (config 5) in a loop, for a kernel trace of just that call:
    rocprofv3 --kernel-trace --stats -- python3 tools/prof_inf_sample.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import Diagonal, KFAC, EFB, INF  # noqa: E402

N = 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = models.resnet50().to(dev).train()
diag, kfac = Diagonal(model), KFAC(model)
x = torch.randn(N, 3, 224, 224, device=dev)
logits = model(x)
labels = torch.distributions.Categorical(logits=logits).sample()
torch.nn.functional.cross_entropy(logits, labels).backward()
diag.update(N)
kfac.update(N)
efb = EFB(model, kfac.state)
efb.update(N)
inf = INF(model, diag.state, kfac.state, efb.state, eigvecs=efb.eigvecs)
inf.update(rank=100)
inf.invert(1.0, 1000.0)
for _ in range(3):
    inf.sample_and_replace()
torch.cuda.synchronize()
print("MARK", flush=True)
t0 = time.perf_counter()
for _ in range(20):
    inf.sample_and_replace()
torch.cuda.synchronize()
print(f"inf.sample_and_replace: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
