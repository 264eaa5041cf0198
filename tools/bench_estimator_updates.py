#!/usr/bin/env python
"""Steady-state time of the per-batch calls of config 5's chain on ResNet-50 (N = 32): Diagonal / KFAC / EFB update."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import Diagonal, KFAC, EFB  # noqa: E402


def timed(name, fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms")


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.resnet50().to(dev).train()
    diag, kfac = Diagonal(model), KFAC(model)
    x = torch.randn(32, 3, 224, 224, device=dev)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    kfac.update(32)
    efb = EFB(model, kfac.state)
    timed("diag.update", lambda: diag.update(32))
    timed("kfac.update", lambda: kfac.update(32))
    timed("efb.update", lambda: efb.update(32))
    diag.invert(1.0, 1000.0)
    efb.invert(1.0, 1000.0)
    timed("diag.invert", lambda: diag.invert(1.0, 1000.0))
    timed("efb.invert", lambda: efb.invert(1.0, 1000.0))
    timed("diag.sample_and_replace", diag.sample_and_replace)
    timed("efb.sample_and_replace", efb.sample_and_replace)


if __name__ == "__main__":
    main()
