#!/usr/bin/env python
"""Full Diagonal -> KFAC -> EFB -> INF chain on a model (the order of tutorial.ipynb cells 9-17 of the
reference), timing every stage on the GPU.   python tools/run_chain.py --model resnet18 --batch 8"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import Diagonal, EFB, INF, KFAC  # noqa: E402


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {1e3 * (time.perf_counter() - t0):10.1f} ms", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet18")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rank", type=int, default=100)
    ap.add_argument("--size", type=int, default=224)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    spec = {"lenet5": (models.lenet5, (1, 28, 28)), "resnet18": (models.resnet18, (3, args.size, args.size)),
            "resnet50": (models.resnet50, (3, args.size, args.size))}[args.model]
    model = spec[0]().to(dev).train()
    x = torch.randn(args.batch, *spec[1], device=dev)

    def fwd_bwd():
        logits = model(x)
        labels = torch.distributions.Categorical(logits=logits.detach()).sample()
        loss = torch.nn.functional.cross_entropy(logits, labels)
        model.zero_grad()
        loss.backward()

    kfac, diag = KFAC(model), Diagonal(model)
    timed("forward+backward", fwd_bwd)
    timed("KFAC.update", lambda: kfac.update(args.batch))
    timed("Diagonal.update", lambda: diag.update(args.batch))
    timed("KFAC.invert(1, 1000)", lambda: kfac.invert(1.0, 1000.0))
    timed("KFAC.sample_and_replace", kfac.sample_and_replace)
    model.load_state_dict(kfac.model_state)
    efb = timed("EFB ctor (eigenvectors)", lambda: EFB(model, kfac.state))
    from curvature_amd import ops
    print("   block-Jacobi sweeps:", getattr(ops.eigh, "last_sweeps", None))
    timed("forward+backward", fwd_bwd)
    timed("EFB.update", lambda: efb.update(args.batch))
    timed("EFB.invert(1, 1000)", lambda: efb.invert(1.0, 1000.0))
    timed("EFB.sample_and_replace", efb.sample_and_replace)
    model.load_state_dict(kfac.model_state)
    inf = INF(model, efb.diags, kfac.state, efb.state)
    inf.eigvecs = efb.eigvecs                  # the reference recomputes them in the ctor (curvatures.py:483)
    timed(f"INF.update(rank={args.rank})", lambda: inf.update(rank=args.rank))
    sizes = [(v[0].shape[1], v[1].shape[1]) for v in inf.state.values()]
    print("   (a, b) per layer:", sizes[:8], "... max ab", max(a * b for a, b in sizes))
    timed("INF.invert(1, 1000)", lambda: inf.invert(1.0, 1000.0))
    timed("INF.sample_and_replace", inf.sample_and_replace)
    ok = all(torch.isfinite(p).all() for p in model.parameters())
    print("finite parameters after INF sample:", bool(ok))


if __name__ == "__main__":
    main()
