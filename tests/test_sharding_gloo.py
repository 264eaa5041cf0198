"""The N > 1 path on CPU: LPT layer partition and the single all-gather of sampled parameters, run
with world_size 2 over gloo."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from curvature_amd import sharding


def test_lpt_partition_is_deterministic_and_balanced():
    costs = [10.0, 9.0, 1.0, 8.0, 2.0, 7.0, 3.0, 3.0]
    owner = sharding.lpt_partition(costs, 3)
    assert owner == sharding.lpt_partition(list(costs), 3)
    load = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(3)]
    assert max(load) <= sum(costs) / 3 + max(costs) * 0.5
    assert sharding.lpt_partition(costs, 1) == [0] * len(costs)
    assert sorted(set(sharding.lpt_partition([1.0] * 8, 8))) == list(range(8))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [[(6, 5), (6,)], [(3, 4)], [(10, 2), (10,)], [(1, 1)], [(7, 3), (7,)]]
    costs = [30.0, 12.0, 20.0, 1.0, 21.0]
    shard = sharding.make_shard(costs, rank, world)
    # every rank starts from garbage; the owner fills in the "sampled" values = f(layer index)
    params = [[torch.full(s, -1.0) for s in layer] for layer in shapes]
    for i, layer in enumerate(params):
        if shard.owns(i):
            for j, p in enumerate(layer):
                p.copy_(torch.arange(p.numel(), dtype=torch.float32).view_as(p) + 100 * i + 10 * j)
    shard.allgather_params(params)
    ok = True
    for i, layer in enumerate(params):
        for j, p in enumerate(layer):
            ok &= torch.equal(p, torch.arange(p.numel(), dtype=torch.float32).view_as(p) + 100 * i + 10 * j)
    owned = [i for i in range(len(shapes)) if shard.owns(i)]
    torch.save({"ok": ok, "owned": owned}, os.path.join(result_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_allgather_params_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert all(r["ok"] for r in res)
    assert sorted(res[0]["owned"] + res[1]["owned"]) == [0, 1, 2, 3, 4]      # disjoint cover
    assert res[0]["owned"] and res[1]["owned"]
