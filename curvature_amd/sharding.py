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
    """Stand-alone step time (s) of one layer under the calibrated model of `rank_cost` (factor build + invert + sample).
    Additive use (`make_shard` / `lpt_partition`) ignores that a rank's factors share one inversion sweep; the
    partition of the estimators is `make_layer_shard`, which prices whole ranks."""
    return rank_cost([(n, m, K)])


def lpt_partition(costs: Sequence[float], world: int) -> List[int]:
    """Longest-processing-time-first greedy: owner rank of every item; deterministic on every rank."""
    owner = [0] * len(costs)
    load = [0.0] * world
    for idx in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda k: (load[k], k))
        owner[idx] = r
        load[r] += costs[idx]
    return owner


def rank_cost(dims: Sequence[Sequence[float]], estimator: str = "kfac", rank: int = 100) -> float:
    """Estimated step time (s) of a rank that owns the layers `dims` = [(n, m, K, build_flops), ...] (`build_flops`:
    what the factor build executes for the layer, from the library's own launch plan - `kfac_build_flops`; when it is
    missing the symmetric products (n (n + 1) + m (m + 1)) K are assumed).

    Not additive: the factors of a rank are inverted in one batched sweep, so the serial chains of 64-column steps
    overlap and only the longest one counts, and every phase pays a fixed price for its launches however little work
    they carry (a rank with ten small layers is bound by that, not by flops).  Calibrated on one MI355X with the round-6
    kernels (tools/emulate_sharding.py, profiles/r06_emulate_sharding.txt): a single 4608^2 factor inverts in 2.48 ms =
    72 steps x 34.5 us when the call has at most 64 factors (the library then sweeps a panel's block square in one
    launch), 40 us per step otherwise (the 108 ResNet-50 factors: 6.45 ms); grouped factor build 100 TFLOP/s executed for
    a whole model (5.6 ms for 526 GFLOP) on top of ~0.3 ms of fixed passes; sampling products 0.96 ms for the model's
    156 GFLOP (dense count; half of it is cut away by the triangles), 0.32 ms for one 4608 x 512 layer.

    `estimator`: "kfac" prices update + invert + sample_and_replace of KFAC; "efb" adds the eigendecomposition of the
    rank's factors (HBM-bound block-Jacobi: ~2.8e-12 s per n^3, and never faster than the serial chain of the
    largest factor: ~16 sweeps x n / 32 rounds x 0.19 ms) and EFB's update; "inf" adds INF.invert on top of that:
    fp64 factor-and-invert sweeps and triangular products of the (a b)^2 matrices, a b ~ 0.4 min(n, rank) min(m, rank)
    (ResNet layers at rank 100: 3400-4200), ~7.5e-14 s per (a b)^3, again bounded below by the largest one's chain."""
    if not dims:
        return 0.0
    flops = sum(d[3] if len(d) > 3 else (d[0] * (d[0] + 1.0) + d[1] * (d[1] + 1.0)) * d[2] for d in dims)
    build = 0.25e-3 + 0.035e-3 * min(len(dims), 10) + flops / 100e12
    sample = 0.2e-3 + sum(2.0 * (d[0] * d[0] * d[1] + d[0] * d[1] * d[1]) for d in dims) / 200e12 + 2e-6 * len(dims)
    chain = max(max(d[0], d[1]) for d in dims) / 64.0 * (34.5e-6 if 2 * len(dims) <= 64 else 40e-6)
    # (the triangular inverse runs in fp32 off the chain, the far updates through LDS-DMA: 6.45 ms for the 108 ResNet-50
    # factors, 2.48 ms for one 4608^2)
    invert = 0.2e-3 + 0.7 * chain + sum((2.0 / 3.0) * (d[0] ** 3 + d[1] ** 3) for d in dims) / 75e12 + 6e-6 * len(dims)
    total = build + invert + sample
    if estimator in ("efb", "inf"):
        n3 = sum(float(d[0]) ** 3 + float(d[1]) ** 3 for d in dims)
        eig_chain = 16.0 * (max(max(d[0], d[1]) for d in dims) / 32.0) * 0.19e-3
        total += max(2.8e-12 * n3, eig_chain)
        total += 0.2e-3 + sum(2.0 * (d[1] * d[1] * d[0] + d[1] * d[0] * d[0]) for d in dims) / 60e12      # EFB.update
    if estimator == "inf":
        ab = [max(float(rank), 0.4 * min(d[0], rank) * min(d[1], rank)) for d in dims]
        ab = [min(x, float(d[0]) * d[1]) for x, d in zip(ab, dims)]
        total += max(7.5e-14 * sum(x ** 3 for x in ab), max(ab) / 64.0 * 2 * 48e-6)
    return total


def layer_geometry(layer, x_shape: Sequence[int], g_shape: Sequence[int]):
    """The two factor geometries (A side, G side) of a Linear / Conv2d layer as dicts of curv_factor_desc fields, from
    the shapes of its recorded input and grad_output."""
    bias = int(layer.bias is not None)
    if layer.__class__.__name__ == "Conv2d":
        N, C, H, W = x_shape
        (kh, kw), (sh, sw), (ph, pw) = layer.kernel_size, layer.stride, layer.padding
        a = dict(N=N, C=C, H=H, W=W, kh=kh, kw=kw, sh=sh, sw=sw, ph=ph, pw=pw, has_bias=bias)
        g = dict(N=g_shape[0], C=g_shape[1], H=g_shape[2], W=g_shape[3], kh=1, kw=1, sh=1, sw=1, ph=0, pw=0, has_bias=0)
    else:
        rows = 1
        for v in x_shape[:-1]:
            rows *= int(v)
        a = dict(N=rows, C=x_shape[-1], H=1, W=1, kh=1, kw=1, sh=1, sw=1, ph=0, pw=0, has_bias=bias)
        g = dict(N=rows, C=g_shape[-1], H=1, W=1, kh=1, kw=1, sh=1, sw=1, ph=0, pw=0, has_bias=0)
    return a, g


def kfac_build_flops(geometries: Sequence[dict]) -> List[int]:
    """Multiply-add flops the library's launch plan executes for each factor geometry (curv_kfac_plan_info, host
    only - no GPU needed): dim (dim + 1) K for a symmetric product, the sum over its 29 shifted correlations for a 3x3 /
    stride 1 / padding 1 factor.  The partition asks the planner instead of re-stating its eligibility rules."""
    import ctypes
    from . import _lib
    n = len(geometries)
    if n == 0:
        return []
    arr = (_lib.curv_factor_desc * n)()
    for d, a in zip(geometries, arr):
        for k, v in d.items():
            setattr(a, k, int(v))
        a.scale = 1.0
    fields = 25                                           # CURV_PLAN_INFO_FIELDS
    out = (ctypes.c_longlong * (fields * n))()
    _lib.check(_lib.lib().curv_kfac_plan_info(arr, n, out), "curv_kfac_plan_info")
    return [int(out[fields * i + fields - 1]) for i in range(n)]


def layer_dims(layers, shapes) -> List[tuple]:
    """[(n, m, K, build flops)] per layer for `partition_layers` / `make_layer_shard`; `shapes[layer]` = (input
    shape, grad_output shape) of one batch."""
    geoms = []
    for layer in layers:
        a, g = layer_geometry(layer, *shapes[layer])
        geoms += [a, g]
    flops = kfac_build_flops(geoms)
    dims = []
    for i, layer in enumerate(layers):
        a, g = geoms[2 * i], geoms[2 * i + 1]
        n = a["C"] * a["kh"] * a["kw"] + a["has_bias"]
        if layer.__class__.__name__ == "Conv2d":
            (kh, kw), (sh, sw), (ph, pw) = layer.kernel_size, layer.stride, layer.padding
            K = a["N"] * ((a["H"] + 2 * ph - kh) // sh + 1) * ((a["W"] + 2 * pw - kw) // sw + 1)
        else:
            K = a["N"]
        dims.append((n, g["C"], K, float(flops[2 * i] + flops[2 * i + 1])))
    return dims


def partition_layers(dims: Sequence[Sequence[int]], world: int, estimator: str = "kfac", rank: int = 100) -> List[int]:
    """Greedy partition under `rank_cost`: layers in descending stand-alone cost, each to the rank whose
    estimated step time grows the least past the current maximum.  Deterministic on every rank."""
    order = sorted(range(len(dims)), key=lambda i: (-rank_cost([dims[i]], estimator, rank), i))
    groups: List[List[int]] = [[] for _ in range(world)]
    owner = [0] * len(dims)
    for idx in order:
        best, best_key = 0, None
        for r in range(world):
            c = rank_cost([dims[i] for i in groups[r]] + [dims[idx]], estimator, rank)
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
            # one flat vector, grouped by owning rank: rank r's parameters are flat[displs[r] : displs[r] + sizes[r]]
            sizes = [0] * self.world
            for i, ps in enumerate(params_per_layer):
                sizes[owner[i]] += sum(p.numel() for p in ps)
            displs = [0] * self.world
            for r in range(1, self.world):
                displs[r] = displs[r - 1] + sizes[r - 1]
            flat = torch.zeros(max(sum(sizes), 1), dtype=ref.dtype, device=ref.device)
            pack, unpack = [], []
            cursor = list(displs)
            for i, ps in enumerate(params_per_layer):
                r = owner[i]
                for p in ps:
                    if not p.is_contiguous():
                        raise RuntimeError("sharded parameters must be contiguous")
                    n = p.numel()
                    seg = flat[cursor[r]:cursor[r] + n]
                    if r == self.rank:
                        pack.append((seg, p.detach().reshape(-1)))
                    else:
                        unpack.append((p.detach().reshape(-1), seg))
                    cursor[r] += n
            cache = {"key": key, "sizes": sizes, "displs": displs, "flat": flat, "pack": pack, "unpack": unpack,
                     "pack_plan": None, "unpack_plan": None}
            if ref.is_cuda:
                from . import ops
                cache["pack_plan"] = ops.CopyPlan([d for d, _ in pack], [s_ for _, s_ in pack])
                cache["unpack_plan"] = ops.CopyPlan([d for d, _ in unpack], [s_ for _, s_ in unpack])
            while len(plans) >= 2:
                plans.pop(next(iter(plans)))
            plans[key] = cache
        flat, sizes, displs = cache["flat"], cache["sizes"], cache["displs"]
        if cache["pack_plan"] is not None:
            cache["pack_plan"].run()
        else:
            for d, s_ in cache["pack"]:
                d.copy_(s_)
        backend = dist.get_backend(self.group)
        if flat.is_cuda and backend == "nccl" and flat.dtype == torch.float32 and self._rccl_ready(flat.device):
            # opt-in (CURV_RCCL_ALLGATHER=1): variable-count all-gather in place over RCCL (curv_allgather_weights), each
            # segment travelling once - the shards differ several-fold in size, nothing is padded to the largest
            self._allgather_rccl(flat, sizes, displs)
        elif flat.is_cuda and backend == "nccl":
            # the default: the one collective is torch's all_gather_into_tensor (RCCL) on equal-sized, padded shards.
            # The library's own communicator stays opt-in until a run with >= 2 RCCL ranks has compared `flat` bit for
            # bit between the two paths (no such run exists: this pool has one GPU per box)
            cap = max(max(sizes), 1)
            pad = cache.get("pad")
            if pad is None:
                pad = cache["pad"] = (torch.zeros(cap, dtype=flat.dtype, device=flat.device),
                                      torch.empty(self.world * cap, dtype=flat.dtype, device=flat.device))
            mine, gathered = pad
            mine[:sizes[self.rank]].copy_(flat[displs[self.rank]:displs[self.rank] + sizes[self.rank]])
            dist.all_gather_into_tensor(gathered, mine, group=self.group)
            for r in range(self.world):
                if r != self.rank and sizes[r]:
                    flat[displs[r]:displs[r] + sizes[r]].copy_(gathered[r * cap:r * cap + sizes[r]])
        else:
            # gloo (the CPU tests, and test configurations with several ranks on one GPU): equal-sized staging buffers
            cap = max(max(sizes), 1)
            mine = torch.zeros(cap, dtype=ref.dtype)
            mine[:sizes[self.rank]] = flat[displs[self.rank]:displs[self.rank] + sizes[self.rank]].cpu()
            parts = [torch.empty(cap, dtype=ref.dtype) for _ in range(self.world)]
            dist.all_gather(parts, mine, group=self.group)
            for r in range(self.world):
                if r != self.rank and sizes[r]:
                    flat[displs[r]:displs[r] + sizes[r]].copy_(parts[r][:sizes[r]])
        if cache["unpack_plan"] is not None:
            cache["unpack_plan"].run()
        else:
            for d, s_ in cache["unpack"]:
                d.copy_(s_)

    # ------------------------------------------------------------------ RCCL communicator of this shard
    def _rccl_ready(self, device: torch.device) -> bool:
        """True when the library's own RCCL communicator was asked for (CURV_RCCL_ALLGATHER=1; the default is torch's
        all_gather_into_tensor) and every rank of the shard has it (built at the first call).  The decision is collective -
        one all-reduce(MIN) of a success flag - so that all ranks take the same branch."""
        ready = self.__dict__.get("_rccl_ok")
        if ready is None:
            import os
            # the environment is read per process and a multi-node launcher need not propagate it: a rank without the
            # variable still takes part in the all-reduce below (with ok = 0), so that every rank takes the same branch
            ok = int(os.environ.get("CURV_RCCL_ALLGATHER", "0") not in ("", "0"))
            if ok:
                # can THIS rank reach RCCL through the library at all (dlopen, symbols)?  A local probe without side effects
                # (curv_rccl_available: no ncclGetUniqueId, whose bootstrap listener would stay behind on every rank): the
                # collective steps below must not start unless every rank can take part in them
                from . import _lib
                try:
                    ok = int(_lib.lib().curv_rccl_available() == 1)
                except Exception:                                         # noqa: BLE001
                    ok = 0
                if not ok:
                    import warnings
                    warnings.warn("curv_allgather_weights is not available on this rank (RCCL could not be bound); using "
                                  "torch.distributed.all_gather_into_tensor")
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            ready = bool(int(flag.item()))
            if ready:
                self._rccl_comm(device)                                   # collective: id broadcast + ncclCommInitRank
            self.__dict__["_rccl_ok"] = ready
        return ready

    def _rccl_comm(self, device: torch.device):
        """An RCCL communicator of this shard's ranks for `curv_allgather_weights` (torch.distributed does not hand out
        its own ncclComm_t): rank 0 draws the id, one torch.distributed broadcast ships its 128 bytes."""
        comm = self.__dict__.get("_comm")
        if comm is None:
            import ctypes
            from . import _lib
            L = _lib.lib()
            ident = (ctypes.c_ubyte * 128)()
            if self.rank == 0:
                _lib.check(L.curv_comm_unique_id(ident), "curv_comm_unique_id")
            box = torch.tensor(list(ident), dtype=torch.uint8, device=device)
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast(box, src=src, group=self.group)
            ident = (ctypes.c_ubyte * 128)(*box.cpu().tolist())
            handle = ctypes.c_void_p()
            with torch.cuda.device(device):
                _lib.check(L.curv_comm_init(ctypes.byref(handle), self.world, ident, self.rank), "curv_comm_init")
            comm = self.__dict__["_comm"] = handle
        return comm

    def _allgather_rccl(self, flat: torch.Tensor, sizes: Sequence[int], displs: Sequence[int]) -> None:
        import ctypes
        from . import _lib
        comm = self._rccl_comm(flat.device)
        counts = (ctypes.c_longlong * self.world)(*sizes)
        offs = (ctypes.c_longlong * self.world)(*displs)
        with torch.cuda.device(flat.device):
            _lib.check(_lib.lib().curv_allgather_weights(comm, _lib.stream_ptr(), flat.data_ptr(), counts, offs),
                       "curv_allgather_weights")

    def rccl_ranks(self) -> int:
        """Ranks of the RCCL communicator the all-gather runs on (0 before its first use / with another backend)."""
        return self.world if self.__dict__.get("_comm") is not None else 0

    def close(self) -> None:
        """Destroy the RCCL communicator (optional; collective: call it on every rank)."""
        comm = self.__dict__.pop("_comm", None)
        if comm is not None:
            from . import _lib
            _lib.check(_lib.lib().curv_comm_destroy(comm), "curv_comm_destroy")


def make_shard(costs: Sequence[float], rank: Optional[int] = None, world: Optional[int] = None, group=None) -> Shard:
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    return Shard(lpt_partition(costs, world), rank, world, group)


def make_layer_shard(dims: Sequence[Sequence[int]], rank: Optional[int] = None, world: Optional[int] = None,
                     group=None, estimator: str = "kfac", inf_rank: int = 100) -> Shard:
    """Shard from the layer sizes [(n, m, K, build flops), ...] (`layer_dims`) with the calibrated, non-additive rank
    cost model; `estimator` = "kfac" | "efb" | "inf": which chain the step consists of."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    return Shard(partition_layers(dims, world, estimator, inf_rank), rank, world, group)
