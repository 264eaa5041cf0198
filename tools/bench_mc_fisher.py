#!/usr/bin/env python
"""MC-Fisher driver (curvature_amd.factors.compute_factors) on ResNet-50: time per batch with the A side
rebuilt per label draw (the reference's loop) and built once per forward pass."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.factors import compute_factors  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    samples, N, nb = 4, 32, 3
    torch.manual_seed(0)
    model = models.resnet50().to(dev)
    data = [(torch.randn(N, 3, 224, 224, device=dev), None) for _ in range(nb)]
    for share in (False, True):
        compute_factors(None, model, data[:1], estimator="kfac", samples=samples, device=dev, share_inputs=share)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        compute_factors(None, model, data, estimator="kfac", samples=samples, device=dev, share_inputs=share)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nb
        print(f"share_inputs={share}: {dt * 1e3:.1f} ms per batch (N={N}, {samples} label draws; forward + "
              f"{samples} backward + factor updates)")


if __name__ == "__main__":
    main()
