#!/usr/bin/env python
"""What layer sharding gives on N GPUs, measured on ONE: each rank's share of the ResNet-50 step (update +
invert + sample of the layers the LPT partition assigns to it) is run in turn; the N-GPU step time is the
slowest rank plus the all-gather (not included here)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, sharding  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.resnet50().to(dev).train()
    rows = models.layer_table(models.resnet50(), (3, 224, 224))
    x = torch.randn(32, 3, 224, 224, device=dev)
    for world in (1, 2, 4, 8):
        mods = dict(model.named_modules())
        dims = [(r["n"], r["m"], 32 * r["L"], sharding.conv_build_flops(r["n"], r["m"], 32 * r["L"], mods[r["name"]], 32))
                for r in rows]
        costs = [sharding.rank_cost([d]) for d in dims]
        owner = sharding.partition_layers(dims, world)
        est = [sharding.rank_cost([d for d, o in zip(dims, owner) if o == r]) * 1e3 for r in range(world)]
        print("   model ms per rank:", " ".join(f"{e:.1f}" for e in est))
        times = []
        for rank in range(world):
            kfac = KFAC(model)
            kfac.shard = sharding.Shard(owner, rank, world)
            kfac._allgather_sampled = lambda: None
            logits = model(x)
            labels = torch.distributions.Categorical(logits=logits).sample()
            model.zero_grad()
            torch.nn.functional.cross_entropy(logits, labels).backward()

            def step():
                kfac.update(32)
                kfac.invert(1.0, 1000.0)
                kfac.sample_and_replace()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / 4 * 1e3)
            if world == 8:
                # per-phase times of this rank (each phase synchronised: an upper bound of its share of the step)
                ph = []
                for fn in (lambda: kfac.update(32), lambda: kfac.invert(1.0, 1000.0), kfac.sample_and_replace):
                    fn()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(4):
                        fn()
                    torch.cuda.synchronize()
                    ph.append((time.perf_counter() - t1) / 4 * 1e3)
                own = [i for i, o in enumerate(owner) if o == rank]
                print(f"      rank {rank}: layers {own} dims {[tuple(dims[i][:2]) for i in own]}: update {ph[0]:.2f} invert {ph[1]:.2f} sample {ph[2]:.2f} ms")
            for h in kfac.hooks:
                h.remove()
        mx = max(times)
        print(f"world {world}: per-rank ms " + " ".join(f"{t:.1f}" for t in times) +
              f" -> step {mx:.1f} ms (+ all-gather), model cost balance {max(sum(c for c, o in zip(costs, owner) if o == r) for r in range(world)) / (sum(costs) / world):.2f}")


if __name__ == "__main__":
    main()
