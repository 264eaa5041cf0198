#!/usr/bin/env python
"""SURVEY 8(d) configs 2 and 3 on one GPU: KFAC update / invert / sample_and_replace step times for LeNet-5
(N = 100, invert(0.5, 1) as scripts/test.py) and ImageNet ResNet-18 (N = 32, invert(1, 1000)).  Not the
headline bench (that is bench.py on ResNet-50); LeNet is launch-latency-bound by construction."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402


def run(name, model, x, hyper, iters=20):
    dev = x.device
    kfac = KFAC(model)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    N = x.shape[0]
    phases = {"update": lambda: kfac.update(N), "invert": lambda: kfac.invert(*hyper), "sample": kfac.sample_and_replace}
    for fn in phases.values():
        fn()
    out = {}
    for key, fn in phases.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        out[key] = (time.perf_counter() - t0) / iters
    step = sum(out.values())
    L = len(kfac.state)
    print(f"{name}: {L} layers, N={N}: update {out['update'] * 1e3:.3f} ms, invert {out['invert'] * 1e3:.3f} ms, "
          f"sample_and_replace {out['sample'] * 1e3:.3f} ms -> step {step * 1e3:.3f} ms = {L / step:.0f} layers/s")


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    run("LeNet-5", models.lenet5().to(dev).train(), torch.rand(100, 1, 28, 28, device=dev), (0.5, 1.0))
    run("ResNet-18", models.resnet18().to(dev).train(), torch.randn(32, 3, 224, 224, device=dev), (1.0, 1000.0))


if __name__ == "__main__":
    main()
