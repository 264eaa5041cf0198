"""Layer sharding across the GPUs of one node.

Per-layer independence is the reference's own modelling assumption (curvature/curvatures.py:20-21):
``update``, ``invert`` and ``sample`` touch only ``state[layer]``.  So each rank owns a disjoint group of
layers (static LPT partition by estimated cost), runs the replicated forward/backward to have the
activations/gradients of its layers locally, and the ONLY data-path collective is one all-gather of the
sampled parameters per ``sample_and_replace()`` (RCCL over xGMI through ``torch.distributed``, backend
"nccl"; "gloo" in the CPU tests).
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def layer_cost(n: int, m: int, K: int) -> float:
    """Rough per-layer time model (arbitrary units): factor build + invert + sample at measured rates."""
    build = (n * (n + 1.0) + m * (m + 1.0)) * K / 60e12        # executed SYRK flops at ~60 TFLOP/s
    invert = (2.0 / 3.0) * (n ** 3 + m ** 3) / 2e12 + (n + m) / 64 * 3 * 25e-6 / 8   # fp64 sweeps, launch-bound part
    sample = (2.0 * n * n * m + 2.0 * n * m * m) / 20e12
    return build + invert + sample


def lpt_partition(costs: Sequence[float], world: int) -> List[int]:
    """Longest-processing-time-first greedy: owner rank of every item; deterministic on every rank."""
    owner = [0] * len(costs)
    load = [0.0] * world
    for idx in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[idx] = r
        load[r] += costs[idx]
    return owner


class Shard:
    """Which layers this rank owns, and the all-gather that reassembles sampled parameters."""

    def __init__(self, owner: Sequence[int], rank: int, world: int, group=None):
        self.owner, self.rank, self.world, self.group = list(owner), rank, world, group

    def owns(self, index: int) -> bool:
        return self.owner[index] == self.rank

    def allgather_params(self, params_per_layer: List[List[torch.Tensor]]) -> None:
        """params_per_layer[i] = parameter tensors of layer i (same shapes on every rank).  After the call
        every rank holds the owner's values for every layer.  One all-gather of equal-sized packed shards."""
        if self.world == 1:
            return
        sizes = [0] * self.world
        for i, ps in enumerate(params_per_layer):
            sizes[self.owner[i]] += sum(p.numel() for p in ps)
        cap = max(max(sizes), 1)
        ref = params_per_layer[0][0]
        mine = torch.zeros(cap, dtype=ref.dtype, device=ref.device)
        pos = 0
        for i, ps in enumerate(params_per_layer):
            if self.owner[i] == self.rank:
                for p in ps:
                    mine[pos:pos + p.numel()].copy_(p.detach().reshape(-1))
                    pos += p.numel()
        gathered = torch.empty(self.world * cap, dtype=ref.dtype, device=ref.device)
        if mine.is_cuda and dist.get_backend(self.group) == "gloo":
            # test configurations only (several ranks sharing one GPU over gloo): stage through the host
            host = torch.empty(self.world * cap, dtype=ref.dtype)
            dist.all_gather(list(host.chunk(self.world)), mine.cpu(), group=self.group)
            gathered.copy_(host)
        else:
            dist.all_gather(list(gathered.chunk(self.world)), mine, group=self.group)   # the one collective
        cursor = [r * cap for r in range(self.world)]
        for i, ps in enumerate(params_per_layer):
            r = self.owner[i]
            for p in ps:
                if r != self.rank:
                    p.detach().copy_(gathered[cursor[r]:cursor[r] + p.numel()].view_as(p))
                cursor[r] += p.numel()


def make_shard(costs: Sequence[float], rank: Optional[int] = None, world: Optional[int] = None, group=None) -> Shard:
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    return Shard(lpt_partition(costs, world), rank, world, group)
