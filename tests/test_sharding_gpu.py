"""The N > 1 path with real estimators: two rank processes (gloo rendezvous, both on GPU 0 - the box has one GPU)
run the sharded Diagonal -> KFAC -> EFB -> INF chain on LeNet-5 and ONE all-gather per sample_and_replace()
reassembles the sampled parameters; every rank must end up with exactly the unsharded run's weights
(reference independence of layers: curvature/curvatures.py:20-21, 414-436, 487-530)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIMS = [(26, 6, 6272), (151, 16, 800), (401, 120, 8), (121, 84, 8), (85, 10, 8)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _chain(rank, world, port, out_dir):
    from curvature_amd import models, sharding
    from curvature_amd.curvatures import KFAC, Diagonal, EFB, INF
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    g1 = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "g1_kfac_lenet.npz")).items()}
    model = models.lenet5()
    layers = [m for m in model.modules() if m.__class__.__name__ in ("Conv2d", "Linear")]
    with torch.no_grad():
        for li, layer in enumerate(layers):
            layer.weight.copy_(g1[f"w_l{li}"])
            layer.bias.copy_(g1[f"bias_l{li}"])
    model = model.to(dev).eval()
    shard = sharding.make_shard([sharding.layer_cost(*d) for d in DIMS], rank, world) if world > 1 else None
    kfac, diag = KFAC(model, shard=shard), Diagonal(model, shard=shard)

    def replay(b, ests):
        # recorded inputs / gradients of golden g1 instead of a backward pass per process: MIOpen's weight-gradient
        # kernels are not bitwise reproducible, and this test compares processes bit for bit
        for li, layer in enumerate(layers):
            layer.weight.grad = g1[f"b{b}_l{li}_gw"].to(dev)
            layer.bias.grad = g1[f"b{b}_l{li}_gb"].to(dev)
        for est in ests:
            if isinstance(est, KFAC):
                for li, layer in enumerate(layers):
                    est.record[layer] = [g1[f"b{b}_l{li}_x"].to(dev), g1[f"b{b}_l{li}_g"].to(dev)]
            est.update(8)

    for b in range(2):
        replay(b, [kfac, diag])
    efb = EFB(model, kfac.state, shard=shard)
    replay(0, [efb])
    inf = INF(model, diag.state, kfac.state, efb.state, shard=shard, eigvecs=efb.eigvecs)
    inf.update(rank=10)
    result = {}
    for name, est in (("kfac", kfac), ("efb", efb), ("inf", inf)):
        est.invert(add=[0.5, 1.0, 2.0, 0.25, 3.0], multiply=[1.0, 10.0, 100.0, 5.0, 50.0])
        sizes = [(n, m) for n, m, _ in DIMS]
        if name == "inf":
            noise = {l: torch.randn(n * m, generator=torch.Generator().manual_seed(li)).to(dev)
                     for li, (l, (n, m)) in enumerate(zip(layers, sizes))}
        else:
            noise = {l: torch.randn(n, m, generator=torch.Generator().manual_seed(li)).to(dev)
                     for li, (l, (n, m)) in enumerate(zip(layers, sizes))}
        noise = {l: z for l, z in noise.items() if shard is None or shard.owns(layers.index(l))}
        est.sample_and_replace(noise=noise)
        result[name] = [(l.weight.detach().cpu().clone(), l.bias.detach().cpu().clone()) for l in layers]
        result[name + "_owned"] = [layers.index(l) for l in est.state.keys()]
    torch.save(result, os.path.join(out_dir, f"w{world}_r{rank}.pt"))
    if world > 1:
        dist.destroy_process_group()


def test_sharded_chain_world2_equals_unsharded(tmp_path):
    assert torch.cuda.is_available()
    mp.spawn(_chain, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_chain, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    full = torch.load(os.path.join(tmp_path, "w1_r0.pt"))
    parts = [torch.load(os.path.join(tmp_path, f"w2_r{r}.pt")) for r in range(2)]
    for name in ("kfac", "efb", "inf"):
        o0, o1 = parts[0][name + "_owned"], parts[1][name + "_owned"]
        assert o0 and o1 and sorted(o0 + o1) == [0, 1, 2, 3, 4], (name, o0, o1)
        for part in parts:                                   # after the all-gather every rank holds every layer
            for (w, b), (wf, bf) in zip(part[name], full[name]):
                assert torch.equal(w, wf) and torch.equal(b, bf), name
