#!/usr/bin/env python
"""Eigensolver workload for the profiler: utils.get_eigenvectors on the 42 KFAC factors of an ImageNet ResNet-18
(N = 8), the constructor work of EFB / INF (tools/collect_profiles.sh runs this under rocprofv3)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, ops  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402
from curvature_amd.utils import get_eigenvectors  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.resnet18().to(dev).train()
    kfac = KFAC(model)
    x = torch.randn(8, 3, 224, 224, device=dev)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    kfac.update(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    get_eigenvectors(kfac.state)
    torch.cuda.synchronize()
    print(f"get_eigenvectors, 42 ResNet-18 factors: {time.perf_counter() - t0:.3f} s, {ops.eigh.last_sweeps} sweeps (largest matrix)")


if __name__ == "__main__":
    main()
