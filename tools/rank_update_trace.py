#!/usr/bin/env python
"""One rank's share of the ResNet-50 step under 8-way layer sharding, in a loop (for rocprofv3 --kernel-trace): which
kernels and gaps make up update() / invert() / sample_and_replace() when a rank owns ~7 layers.  Diagnostics only.
    python tools/rank_update_trace.py [rank] [world]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, sharding  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rank = int(args[0]) if len(args) > 0 else 3
    world = int(args[1]) if len(args) > 1 else 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.resnet50().to(dev).train()
    x = torch.randn(32, 3, 224, 224, device=dev)
    probe = KFAC(model)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    layers = probe._layers()
    dims = sharding.layer_dims(layers, {l: (tuple(probe.record[l][0].shape), tuple(probe.record[l][1].shape)) for l in layers})
    for h in probe.hooks:
        h.remove()
    owner = sharding.partition_layers(dims, world, "kfac")
    kfac = KFAC(model, shard=sharding.Shard(owner, rank, world))
    kfac._allgather_sampled = lambda: None
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    model.zero_grad()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    print("layers", [tuple(dims[i][:2]) for i, o in enumerate(owner) if o == rank])

    def step():
        kfac.update(32)
        kfac.invert(1.0, 1000.0)
        kfac.sample_and_replace()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for name, fn in (("update", lambda: kfac.update(32)), ("invert", lambda: kfac.invert(1.0, 1000.0)),
                     ("sample", kfac.sample_and_replace), ("step", step)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
    if "--host" in sys.argv:
        import cProfile
        import pstats
        for name, fn in (("update", lambda: kfac.update(32)), ("invert", lambda: kfac.invert(1.0, 1000.0, check=False)),
                         ("sample", kfac.sample_and_replace)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f"{name}: host time to enqueue {1e3 * (t1 - t0):.3f} ms")
        pr = cProfile.Profile()
        for _ in range(5):
            torch.cuda.synchronize()
            pr.enable()
            kfac.update(32)
            pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(25)


if __name__ == "__main__":
    main()
