#!/usr/bin/env python
"""What layer sharding gives on N GPUs, measured on ONE: each rank's share of the ResNet-50 step is run in turn; the
N-GPU step time is the slowest rank plus the all-gather of the sampled parameters, which one GPU cannot run: it is
MODELLED (printed separately and labelled as such): every rank receives the other ranks' segments, (W - 1) / W of the
102 MB flat vector, at the all-gather bus bandwidth RCCL reaches over xGMI for messages of this size (7 links x 153 GB/s
peak per GPU; 300 GB/s assumed = 0.28 of link peak, the usual large-message figure) + 20 us per rank's broadcast of the
group call.  No scaling curve has been measured on hardware.

    python tools/emulate_sharding.py            KFAC step: update + invert(1, 1000) + sample_and_replace (config 4)
    python tools/emulate_sharding.py --chain    config 5: EFB constructor (eigenvectors), efb.update, INF.update(100),
                                                inf.invert(1, 1000), inf.sample_and_replace per rank
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, sharding  # noqa: E402
from curvature_amd.curvatures import KFAC, EFB, INF  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    chain = "--chain" in sys.argv
    nocheck = "--nocheck" in sys.argv      # invert(check=False): no host synchronisation inside the step (diagnostic)
    graph = "--graph" in sys.argv          # the rank's step as a replayed HIP graph (curvature_amd.graph) + check()
    estimator = "inf" if chain else "kfac"
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
    for world in ((1, 8) if chain else (1, 2, 4, 8)):
        owner = sharding.partition_layers(dims, world, estimator)
        est = [sharding.rank_cost([d for d, o in zip(dims, owner) if o == r], estimator) * 1e3 for r in range(world)]
        print("   model ms per rank:", " ".join(f"{e:.1f}" for e in est))
        times = []
        for rank in range(world):
            shard = sharding.Shard(owner, rank, world)
            kfac = KFAC(model, shard=shard)
            kfac._allgather_sampled = lambda: None
            model.load_state_dict(kfac.model_state)
            logits = model(x)
            labels = torch.distributions.Categorical(logits=logits).sample()
            model.zero_grad()
            torch.nn.functional.cross_entropy(logits, labels).backward()
            own = [i for i, o in enumerate(owner) if o == rank]
            if not chain:
                def step():
                    kfac.update(32)
                    kfac.invert(1.0, 1000.0, check=not nocheck)
                    kfac.sample_and_replace()
                step()
                step()
                if graph:
                    from curvature_amd.graph import KFACStepGraph
                    g = KFACStepGraph(kfac, add=1.0, multiply=1000.0, batch_size=32)

                    def step():
                        g.replay()
                        g.check()
                    step()
                times.append(timed(step, 4))
                if world == 8:
                    ph = [timed(lambda: kfac.update(32), 4), timed(lambda: kfac.invert(1.0, 1000.0), 4), timed(kfac.sample_and_replace, 4)]
                    print(f"      rank {rank}: layers {own} dims {[tuple(dims[i][:2]) for i in own]}: update {ph[0]:.2f} invert {ph[1]:.2f} "
                          f"sample {ph[2]:.2f} ms")
            else:
                kfac.update(32)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                efb = EFB(model, kfac.state, shard=shard)
                torch.cuda.synchronize()
                t_eig = (time.perf_counter() - t0) * 1e3
                efb._allgather_sampled = lambda: None
                t_upd = timed(lambda: efb.update(32), 2)
                inf = INF(model, efb.diags, kfac.state, efb.state, shard=shard, eigvecs=efb.eigvecs)
                inf._allgather_sampled = lambda: None
                t_iu = timed(lambda: inf.update(rank=100), 1)
                t_ii = timed(lambda: inf.invert(1.0, 1000.0), 2)
                t_is = timed(inf.sample_and_replace, 2)
                times.append(t_eig + t_upd + t_iu + t_ii + t_is)
                print(f"      rank {rank}: {len(own)} layers, largest factor {max(max(dims[i][:2]) for i in own)}: eigenvectors {t_eig:.0f}  "
                      f"efb.update {t_upd:.1f}  inf.update {t_iu:.1f}  inf.invert {t_ii:.1f}  inf.sample_and_replace {t_is:.1f} ms")
                del efb, inf
            for h in kfac.hooks:
                h.remove()
            del kfac
        copies_ms = 0.0
        if world > 1 and not chain:
            # the device copies around the collective, MEASURED (the collective itself cannot run on one GPU): rank 0's
            # batched pack of its own parameters into the flat vector, the padded staging copies of the default
            # (torch all_gather_into_tensor) path, and the batched unpack of the other ranks' segments
            from curvature_amd import ops
            params = [[p.detach().reshape(-1) for p in (l.weight, l.bias) if p is not None] for l in layers]
            sizes = [0] * world
            for i, ps in enumerate(params):
                sizes[owner[i]] += sum(p.numel() for p in ps)
            displs = [sum(sizes[:r]) for r in range(world)]
            flat = torch.zeros(sum(sizes), device=dev)
            cursor, pack, unpack = list(displs), [], []
            for i, ps in enumerate(params):
                for p in ps:
                    seg = flat[cursor[owner[i]]:cursor[owner[i]] + p.numel()]
                    (pack if owner[i] == 0 else unpack).append((seg, p) if owner[i] == 0 else (p, seg))
                    cursor[owner[i]] += p.numel()
            pack_plan = ops.CopyPlan([d for d, _ in pack], [s_ for _, s_ in pack])
            unpack_plan = ops.CopyPlan([d for d, _ in unpack], [s_ for _, s_ in unpack])
            cap = max(sizes)
            mine, gathered = torch.zeros(cap, device=dev), torch.empty(world * cap, device=dev)

            def copies():
                pack_plan.run()
                mine[:sizes[0]].copy_(flat[:sizes[0]])
                for r in range(1, world):
                    flat[displs[r]:displs[r] + sizes[r]].copy_(gathered[r * cap:r * cap + sizes[r]])
                unpack_plan.run()
            copies_ms = timed(copies, 10)
        mx = max(times)
        n_params = sum(n * m for n, m, *_ in dims)
        gather_ms = 0.0 if world == 1 else (4.0 * n_params * (world - 1) / world / 300e9 + 20e-6 * world) * 1e3
        print(f"world {world}: per-rank ms " + " ".join(f"{t:.1f}" for t in times) + f" -> slowest rank {mx:.1f} ms + MODELLED "
              f"all-gather {gather_ms:.2f} ms ({4.0 * n_params / 1e6:.0f} MB vector, variable counts, unpadded) + MEASURED pack / "
              f"staging / unpack copies {copies_ms:.2f} ms = {mx + gather_ms + copies_ms:.1f} ms")


if __name__ == "__main__":
    main()
