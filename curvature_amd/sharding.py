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
    build = (n * (n + 1.0) + m * (m + 1.0)) * K / 80e12        # executed SYRK flops at ~80 TFLOP/s
    # invert: throughput part at ~20 TFLOP/s (fp64) + the serial chain of 64-column steps (~80 us each; the
    # chains of a rank's factors run side by side, so only a fraction of it adds up)
    invert = (2.0 / 3.0) * (n ** 3 + m ** 3) / 20e12 + (n + m) / 64 * 80e-6 * 0.3
    sample = (n * n * m + n * m * m) / 30e12                    # triangular GEMMs at ~30 TFLOP/s
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


def rank_cost(dims: Sequence[Sequence[float]]) -> float:
    """Estimated step time (s) of a rank that owns the layers `dims` = [(n, m, K) or (n, m, K, build_flops), ...].
    Not additive: the factors of a rank are inverted in one batched sweep, so the serial chains of 64-column steps
    overlap and only the longest one counts (calibrated on MI355X: single 4608^2 factor 3.9 ms = 72 steps x 54 us,
    three of them 6.6 ms, all 108 ResNet-50 factors 9.6 ms; build 85 TFLOP/s executed, sampling GEMMs 95 TFLOP/s).
    `build_flops`: what the factor build of the layer executes when that is not (n (n + 1) + m (m + 1)) K - a 3x3 /
    stride 1 A factor assembled from shifted correlations costs a third of it (`conv_build_flops`)."""
    if not dims:
        return 0.0
    build = sum(d[3] if len(d) > 3 else (d[0] * (d[0] + 1.0) + d[1] * (d[1] + 1.0)) * d[2] for d in dims) / 85e12
    sample = sum(2.0 * (d[0] * d[0] * d[1] + d[0] * d[1] * d[1]) for d in dims) / 95e12
    chain = max(max(d[0], d[1]) for d in dims) / 64.0 * 54e-6
    invert = 0.7 * chain + sum((2.0 / 3.0) * (d[0] ** 3 + d[1] ** 3) for d in dims) / 42e12
    return build + invert + sample + 1.0e-3


def conv_build_flops(n: int, m: int, K: int, layer=None, batch: int = 0) -> float:
    """Multiply-add flops the factor build executes for a layer (both factors).  The library assembles the A factor
    of a 3x3 / stride 1 / padding 1 convolution without bias whose channel count is a multiple of 128 from 13 shifted
    correlations + border strips (csrc/syrk_corr.hip; needs >= 8 samples): about 13 / 40.5 of the symmetric product,
    more on small images because of the two zero columns per row (curv_kfac_plan_info gives the exact figure)."""
    direct_a, direct_g = n * (n + 1.0) * K, m * (m + 1.0) * K
    if layer is not None and layer.__class__.__name__ == "Conv2d" and tuple(layer.kernel_size) == (3, 3) and \
            tuple(layer.stride) == (1, 1) and tuple(layer.padding) == (1, 1) and layer.bias is None and \
            layer.in_channels % 128 == 0 and batch >= 8:
        return direct_a * 0.36 + direct_g
    return direct_a + direct_g


def partition_layers(dims: Sequence[Sequence[int]], world: int) -> List[int]:
    """Greedy partition under `rank_cost`: layers in descending stand-alone cost, each to the rank whose
    estimated step time grows the least past the current maximum.  Deterministic on every rank."""
    order = sorted(range(len(dims)), key=lambda i: (-rank_cost([dims[i]]), i))
    groups: List[List[int]] = [[] for _ in range(world)]
    owner = [0] * len(dims)
    for idx in order:
        best, best_key = 0, None
        for r in range(world):
            c = rank_cost([dims[i] for i in groups[r]] + [dims[idx]])
            key = (c, r)
            if best_key is None or key < best_key:
                best, best_key = r, key
        groups[best].append(idx)
        owner[idx] = best
    return owner


class Shard:
    """Which layers this rank owns, and the all-gather that reassembles sampled parameters."""

    def __init__(self, owner: Sequence[int], rank: int, world: int, group=None, *, force_collective: bool = False):
        self.owner, self.rank, self.world, self.group = list(owner), rank, world, group
        # world == 1 normally skips packing and the collective; `force_collective` runs them anyway (a one-rank
        # process group: the RCCL branch can then be exercised on a box with a single GPU)
        self.force_collective = force_collective

    def owns(self, index: int) -> bool:
        return self.owner[index] == self.rank

    def allgather_params(self, params_per_layer: List[List[torch.Tensor]],
                         owners: Optional[Sequence[int]] = None) -> None:
        """params_per_layer[i] = parameter tensors of layer i (same shapes on every rank).  After the call
        every rank holds the owner's values for every layer.  One all-gather of equal-sized packed shards;
        on the GPU the packing and unpacking are one batched copy each (curv_copy_batched) and the buffers
        and copy plans are kept while the parameter tensors stay where they are.  `owners`: owner rank per
        entry when the list is not the layer list of the partition (Diagonal appends its attention entries)."""
        if self.world == 1 and not self.force_collective:
            return
        owner = list(owners) if owners is not None else self.owner
        if len(owner) != len(params_per_layer):
            raise RuntimeError("allgather_params: one owner per entry expected")
        ref = params_per_layer[0][0]
        key = tuple(p.data_ptr() for ps in params_per_layer for p in ps) + tuple(owner)
        plans = self.__dict__.setdefault("_plans", {})       # two kept: evaluate.eval_bnn alternates two buffer sets
        cache = plans.get(key)
        if cache is None:
            sizes = [0] * self.world
            for i, ps in enumerate(params_per_layer):
                sizes[owner[i]] += sum(p.numel() for p in ps)
            cap = max(max(sizes), 1)
            mine = torch.zeros(cap, dtype=ref.dtype, device=ref.device)
            gathered = torch.empty(self.world * cap, dtype=ref.dtype, device=ref.device)
            pack, unpack = [], []
            cursor = [r * cap for r in range(self.world)]
            pos = 0
            for i, ps in enumerate(params_per_layer):
                r = owner[i]
                for p in ps:
                    if not p.is_contiguous():
                        raise RuntimeError("sharded parameters must be contiguous")
                    n = p.numel()
                    if r == self.rank:
                        pack.append((mine[pos:pos + n], p.detach().reshape(-1)))
                        pos += n
                    else:
                        unpack.append((p.detach().reshape(-1), gathered[cursor[r]:cursor[r] + n]))
                    cursor[r] += n
            cache = {"key": key, "cap": cap, "mine": mine, "gathered": gathered, "pack": pack, "unpack": unpack,
                     "pack_plan": None, "unpack_plan": None}
            if ref.is_cuda:
                from . import ops
                cache["pack_plan"] = ops.CopyPlan([d for d, _ in pack], [s_ for _, s_ in pack])
                cache["unpack_plan"] = ops.CopyPlan([d for d, _ in unpack], [s_ for _, s_ in unpack])
            while len(plans) >= 2:
                plans.pop(next(iter(plans)))
            plans[key] = cache
        mine, gathered = cache["mine"], cache["gathered"]
        if cache["pack_plan"] is not None:
            cache["pack_plan"].run()
        else:
            for d, s_ in cache["pack"]:
                d.copy_(s_)
        backend = dist.get_backend(self.group)
        if mine.is_cuda and backend == "gloo":
            # test configurations only (several ranks sharing one GPU over gloo): stage through the host
            host = torch.empty(self.world * cache["cap"], dtype=ref.dtype)
            dist.all_gather(list(host.chunk(self.world)), mine.cpu(), group=self.group)
            gathered.copy_(host)
        elif backend == "nccl":
            dist.all_gather_into_tensor(gathered, mine, group=self.group)               # the one collective
        else:
            dist.all_gather(list(gathered.chunk(self.world)), mine, group=self.group)
        if cache["unpack_plan"] is not None:
            cache["unpack_plan"].run()
        else:
            for d, s_ in cache["unpack"]:
                d.copy_(s_)


def make_shard(costs: Sequence[float], rank: Optional[int] = None, world: Optional[int] = None, group=None) -> Shard:
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    return Shard(lpt_partition(costs, world), rank, world, group)


def make_layer_shard(dims: Sequence[Sequence[int]], rank: Optional[int] = None, world: Optional[int] = None,
                     group=None) -> Shard:
    """Shard from the layer sizes [(n, m, K), ...] with the calibrated, non-additive rank cost model."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    return Shard(partition_layers(dims, world), rank, world, group)
