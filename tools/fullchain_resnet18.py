#!/usr/bin/env python
"""SURVEY 8(d) config 3/5 at full size: Diagonal -> KFAC -> EFB -> INF(rank=100) -> invert -> sample on an
ImageNet ResNet (random init), with wall-clock per stage."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import Diagonal, KFAC, EFB, INF  # noqa: E402


def stage(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    return out


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = getattr(models, arch)().to(dev).train()
    diag, kfac = Diagonal(model), KFAC(model)
    x = torch.randn(N, 3, 224, 224, device=dev)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    stage("diag.update", lambda: diag.update(N))
    stage("kfac.update", lambda: kfac.update(N))
    efb = stage("EFB ctor (eigenvectors)", lambda: EFB(model, kfac.state))
    stage("efb.update", lambda: efb.update(N))
    inf = stage("INF ctor (eigvecs=efb.eigvecs: no second decomposition)",
                lambda: INF(model, diag.state, kfac.state, efb.state, eigvecs=efb.eigvecs))
    stage("inf.update(100)", lambda: inf.update(rank=100))
    for est, nm in ((diag, "diag"), (kfac, "kfac"), (efb, "efb"), (inf, "inf")):
        stage(f"{nm}.invert(1, 1000)", lambda: est.invert(1.0, 1000.0))
        stage(f"{nm}.invert(1, 1000) again", lambda: est.invert(1.0, 1000.0))
        stage(f"{nm}.sample_and_replace", est.sample_and_replace)
        stage(f"{nm}.sample_and_replace again", est.sample_and_replace)
        ok = all(torch.isfinite(p).all().item() for p in model.parameters())
        print(f"  parameters finite: {ok}")


if __name__ == "__main__":
    main()
