#!/usr/bin/env python
"""Serial vs software-pipelined BNN inference loop (curvature_amd/evaluate.py eval_bnn): ResNet-50 / LeNet-5, KFAC,
a few Monte-Carlo samples, one or several data batches per sample."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402
from curvature_amd.evaluate import eval_bnn  # noqa: E402


def run(name, model, shape, batch, n_batches, samples=20):
    dev = torch.device("cuda:0")
    model = model.to(dev)
    x = torch.randn(batch, *shape, device=dev)
    kfac = KFAC(model)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    kfac.update(batch)
    kfac.invert(1.0, 1000.0)
    data = [(x, torch.zeros(batch, dtype=torch.long)) for _ in range(n_batches)]
    for overlap in (False, True):
        eval_bnn(model, data, kfac, samples=3, device=dev, overlap=overlap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eval_bnn(model, data, kfac, samples=samples, device=dev, overlap=overlap)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / samples * 1e3
        print(f"{name}: batch {batch} x {n_batches} sweeps per sample, overlap={overlap}: {ms:.2f} ms per Monte-Carlo sample")


if __name__ == "__main__":
    torch.manual_seed(0)
    run("LeNet-5", models.lenet5(), (1, 28, 28), 100, 1)
    run("ResNet-50", models.resnet50(), (3, 224, 224), 32, 1)
    run("ResNet-50", models.resnet50(), (3, 224, 224), 32, 4)
    run("ResNet-50", models.resnet50(), (3, 224, 224), 256, 1, samples=10)
