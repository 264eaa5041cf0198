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


def _rccl_world1(rank, port, out_dir, fallback=False):
    """backend "nccl" (= RCCL) with a one-rank group on the box's single GPU: process-group init with a bound
    device, then KFAC.sample_and_replace through Shard.allgather_params' RCCL branch: the library's own export
    curv_allgather_weights (variable-count all-gather in place on the flat parameter vector) on a communicator built by
    curv_comm_unique_id / curv_comm_init."""
    from curvature_amd import models, sharding
    from curvature_amd.curvatures import KFAC
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if fallback:
        os.environ.pop("CURV_RCCL_ALLGATHER", None)               # the default: torch's all_gather_into_tensor on padded shards
    else:
        os.environ["CURV_RCCL_ALLGATHER"] = "1"                   # opt-in: the library's own communicator
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # as bench.py does for N > 1
    assert dist.get_backend() == "nccl"
    torch.manual_seed(0)
    model = models.lenet5().to(dev).eval()
    layers = [m for m in model.modules() if m.__class__.__name__ in ("Conv2d", "Linear")]
    shard = sharding.Shard([0] * len(layers), 0, 1, force_collective=True)
    kfac = KFAC(model, shard=shard)
    x = torch.rand(8, 1, 28, 28, device=dev)
    torch.nn.functional.cross_entropy(model(x), torch.randint(0, 10, (8,), device=dev)).backward()
    kfac.update(8)
    kfac.invert(0.5, 1.0)
    noise = {l: torch.randn(kfac.inv_state[l][0].shape[0], kfac.inv_state[l][1].shape[0], device=dev) for l in layers}
    kfac.sample_and_replace(noise=noise)
    torch.cuda.synchronize()
    got = [p.detach().clone() for l in layers for p in (l.weight, l.bias)]
    # the same sample without any shard: the collective must have left the parameters as the sampler wrote them
    plain = KFAC(model)
    plain.model_state = kfac.model_state
    plain.state, plain.inv_state = kfac.state, kfac.inv_state
    plain.sample_and_replace(noise=noise)
    torch.cuda.synchronize()
    want = [p.detach().clone() for l in layers for p in (l.weight, l.bias)]
    cache = next(iter(shard._plans.values()))
    packed = torch.cat([t.reshape(-1) for t in want])
    ok = all(torch.equal(a, b) for a, b in zip(got, want))
    ok_gather = torch.equal(cache["flat"], packed) and cache["flat"].numel() == sum(cache["sizes"]) and shard.rccl_ranks() == (0 if fallback else 1)
    if fallback:
        torch.save({"ok": ok, "ok_gather": bool(ok_gather), "copy": True, "max": 3.0}, os.path.join(out_dir, "rccl.pt"))
        dist.destroy_process_group()
        return
    # the export on its own: a three-segment vector of which this (only) rank owns everything, counts / displs honoured
    import ctypes
    from curvature_amd import _lib
    vec = torch.arange(1000, dtype=torch.float32, device=dev)
    counts, offs = (ctypes.c_longlong * 1)(1000), (ctypes.c_longlong * 1)(0)
    _lib.check(_lib.lib().curv_allgather_weights(shard._rccl_comm(dev), _lib.stream_ptr(), vec.data_ptr(), counts, offs), "allgather")
    torch.cuda.synchronize()
    ok_gather = ok_gather and torch.equal(vec, torch.arange(1000, dtype=torch.float32, device=dev))
    shard.close()
    # and the collective on its own, with a payload the ranks did not already hold in place
    src = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    dst = torch.zeros_like(src)
    dist.all_gather_into_tensor(dst, src)
    t = torch.tensor([3.0], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                     # bench.py's max-over-ranks timing
    dist.barrier()
    torch.cuda.synchronize()
    torch.save({"ok": ok, "ok_gather": bool(ok_gather), "copy": bool(torch.equal(dst, src)), "max": float(t)},
               os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("fallback", [False, True])
def test_rccl_backend_runs_the_allgather_branch(tmp_path, fallback):
    """the one collective of the path is an RCCL all-gather.  The box has one GPU (RCCL refuses two
    ranks on one device), so the nccl backend is initialised with world_size 1 and `force_collective` sends
    sample_and_replace through pack -> curv_allgather_weights (RCCL group of broadcasts) -> unpack instead of the
    world == 1 early return.  No run with more than one RCCL rank exists: 8-GPU nodes are the driver's.
    `fallback` = the DEFAULT path: torch's all_gather_into_tensor on padded shards; curv_allgather_weights is opt-in
    (CURV_RCCL_ALLGATHER=1) until a run with >= 2 RCCL ranks has compared the two bit for bit.  Same parameters."""
    mp.spawn(_rccl_world1, args=(_free_port(), str(tmp_path), fallback), nprocs=1, join=True)
    res = torch.load(os.path.join(tmp_path, "rccl.pt"))
    assert res == {"ok": True, "ok_gather": True, "copy": True, "max": 3.0}, res


class _AttnNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.inp = torch.nn.Linear(6, 8)
        self.attn = torch.nn.MultiheadAttention(8, 2)
        self.out = torch.nn.Linear(8, 3)

    def forward(self, x):                       # x: (seq, batch, 6)
        h = self.inp(x)
        h, _ = self.attn(h, h, h)
        return self.out(h.mean(dim=0))


def _mha_rank(rank, world, port, out_dir):
    from curvature_amd import sharding
    from curvature_amd.curvatures import Diagonal
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = _AttnNet().to(dev)
    shard = sharding.Shard([0, 1], rank, world)                     # inp -> rank 0, out -> rank 1
    diag = Diagonal(model, shard=shard)
    x = torch.randn(5, 4, 6, device=dev)
    labels = torch.tensor([0, 1, 2, 1], device=dev)
    torch.nn.functional.cross_entropy(model(x), labels).backward()
    diag.update(batch_size=4)
    diag.invert(add=1.0, multiply=10.0)
    mean = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    diag.sample_and_replace()
    torch.cuda.synchronize()
    torch.save({"after": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "mean": mean},
               os.path.join(out_dir, f"mha_r{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_diagonal_attention_entries_agree_across_ranks(tmp_path):
    """Diagonal + nn.MultiheadAttention under a layer shard (curvatures.py:125-129, 159-174): the attention entries
    are outside the layer partition, each rank has its own noise stream - rank 0 draws them and the all-gather
    carries them, so both ranks must end with identical (and actually sampled) attention weights."""
    mp.spawn(_mha_rank, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"mha_r{r}.pt")) for r in range(2))
    for k in r0["after"]:
        assert torch.equal(r0["after"][k], r1["after"][k]), k
        assert not torch.equal(r0["after"][k], r0["mean"][k]), k           # every parameter received a sample


def test_shard_under_the_small_launch_threshold_sums_like_the_unsharded_model(gpu):
    """The factor build has a two-launch form for small launches (csrc/syrk_small.hip, up to CURV_SMALL_MAX_FLOP executed
    flops) that sums in another order than the grouped kernels.  The choice must follow the MODEL, not a rank's share
    (curv_factor_desc.path_hint): here the model is above the threshold and each of the two shares below it, and every
    rank's factors must equal the unsharded run's bit for bit (advisor finding of round 4)."""
    from curvature_amd import _lib, ops, sharding
    from curvature_amd.curvatures import KFAC
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(640, 640), torch.nn.Tanh(), torch.nn.Linear(640, 512), torch.nn.Tanh(),
                                torch.nn.Linear(512, 10)).to(gpu)
    layers = [m for m in model if isinstance(m, torch.nn.Linear)]
    N = 1024
    flop = [ops.small_path_flop(l.in_features + 1, N) + ops.small_path_flop(l.out_features, N) for l in layers]
    owner = [0, 1, 1]
    assert sum(flop) > _lib.SMALL_MAX_FLOP
    for r in range(2):
        assert 0 < sum(f for f, o in zip(flop, owner) if o == r) < _lib.SMALL_MAX_FLOP
    x = torch.randn(N, 640, device=gpu)
    full = KFAC(model)
    parts = [KFAC(model, shard=sharding.Shard(owner, r, 2)) for r in range(2)]
    torch.nn.functional.cross_entropy(model(x), torch.randint(0, 10, (N,), device=gpu)).backward()
    for est in [full] + parts:
        est.update(N)
    torch.cuda.synchronize()
    for li, layer in enumerate(layers):
        mine = parts[owner[li]]
        assert layer in mine.state and layer not in parts[1 - owner[li]].state
        for a, b in zip(full.state[layer], mine.state[layer]):
            assert torch.equal(a, b), li
