#!/usr/bin/env python
"""INF.invert at ResNet-50 size (config 5) in isolation, for a kernel-stats profile of just that call:
    rocprofv3 --kernel-trace --stats -- python3 tools/prof_inf_invert.py
Builds the chain once (not timed), then runs inf.invert(1, 1000) three times between markers printed to stdout."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import Diagonal, KFAC, EFB, INF  # noqa: E402


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    N = 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = getattr(models, arch)().to(dev).train()
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
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        inf.invert(1.0, 1000.0)
        torch.cuda.synchronize()
        print(f"inf.invert: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    sizes = sorted((inf.inv_state[l][3].shape[0] for l in inf.inv_state), reverse=True)
    print("ab sizes:", sizes[:10], "... sum ab^3 =", f"{sum(float(s) ** 3 for s in sizes):.3e}")


if __name__ == "__main__":
    main()
