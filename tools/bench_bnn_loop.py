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
    # weight samples alone: one at a time vs S per launch (sample_many + replace_from), ms per sample
    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    one = timed(kfac.sample_and_replace, 20)
    line = f"{name}: weight samples only: sample_and_replace {one:.3f} ms"
    for S in (2, 4, 8, 16):
        def many():
            bank = kfac.sample_many(S)
            for k in range(S):
                kfac.replace_from(bank, k)
        gen = timed(lambda: kfac.sample_many(S), 5) / S
        line += f" | S={S}: {timed(many, 5) / S:.3f} ms per sample (generation alone {gen:.3f})"
    print(line)
    for S in (8,):
        eval_bnn(model, data, kfac, samples=S, device=dev, samples_per_launch=S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eval_bnn(model, data, kfac, samples=2 * S, device=dev, samples_per_launch=S)
        torch.cuda.synchronize()
        print(f"{name}: batch {batch} x {n_batches}, samples_per_launch={S}: {(time.perf_counter() - t0) / (2 * S) * 1e3:.2f} ms per Monte-Carlo sample")


if __name__ == "__main__":
    torch.manual_seed(0)
    run("LeNet-5", models.lenet5(), (1, 28, 28), 100, 1)
    run("ResNet-50", models.resnet50(), (3, 224, 224), 32, 1)
    run("ResNet-50", models.resnet50(), (3, 224, 224), 32, 4)
    run("ResNet-50", models.resnet50(), (3, 224, 224), 256, 1, samples=10)
