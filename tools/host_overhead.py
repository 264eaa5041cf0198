#!/usr/bin/env python
"""Host-side time of each API call of a KFAC step on ResNet-50 (time until the call returns, GPU idle
before it): what the GPU waits for when nothing else is queued."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.resnet50().to(dev).train()
    kfac = KFAC(model)
    x = torch.randn(32, 3, 224, 224, device=dev)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    for it in range(4):
        out = []
        for name, fn in (("update", lambda: kfac.update(32)), ("invert", lambda: kfac.invert(1.0, 1000.0)),
                         ("sample_and_replace", kfac.sample_and_replace)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out.append(f"{name}: host {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms")
        print(" | ".join(out))


if __name__ == "__main__":
    main()
