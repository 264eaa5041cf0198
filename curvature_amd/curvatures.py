"""MI355X-native estimators behind the reference's plugin API.

Same classes, constructor signatures, method names, public attributes (``state``, ``inv_state``,
``model_state`` ...) and error behaviour as ``curvature/curvatures.py`` of DLR-RM/curvature, so that the
reference's call sequences run unchanged::

    kfac = KFAC(model)
    for images, labels in data:                # scripts/test.py:32-47
        loss = criterion(model(images), sampled_labels); model.zero_grad(); loss.backward()
        kfac.update(batch_size=images.size(0))
    kfac.invert(add=0.5, multiply=1)
    kfac.sample_and_replace()

All arithmetic runs in ``libcurv_hip.so`` (hand-written HIP for gfx950) through the C ABI of
``include/curv_hip.h``; PyTorch only runs the model's forward/backward and owns the tensors.  There is
no CPU fallback: CPU models raise ``RuntimeError``.
"""
import copy
import numbers
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Sequence, Union

import torch
from torch import Tensor
from torch.nn import Module, Sequential

from . import _lib, ops

SUPPORTED_LAYERS = ['Linear', 'Conv2d', 'MultiheadAttention']


def _is_scalar(x) -> bool:
    """Python / numpy real scalars (a superset of what the reference accepts)."""
    return isinstance(x, numbers.Real) or (hasattr(x, "ndim") and getattr(x, "ndim") == 0)


class AttentionProjection:
    """One of the two linear maps of an ``nn.MultiheadAttention`` module seen as a Linear layer: ``'attn_in'`` = the packed
    input projection (in_proj_weight (3E, E), in_proj_bias), ``'attn_out'`` = out_proj (E, E).  KFAC / EFB / INF treat
    each as a layer of its own (the reference raises NotImplementedError for these modules, curvatures.py:303-304,
    351-352, 435-436; SURVEY 8f-4 asks for them): state dicts are keyed by these objects, which are created once per
    module (`of`), so that every estimator of a model uses the same keys.  Self-attention only (query, key and value are
    one tensor: the packed projection is then exactly a Linear(E, 3E) applied to every token)."""

    def __init__(self, module: Module, kind: str):
        self.module, self.kind = module, kind

    @staticmethod
    def of(module: Module):
        cached = module.__dict__.get("_curv_projections")
        if cached is None:
            if not getattr(module, "_qkv_same_embed_dim", True) or module.in_proj_weight is None:
                raise NotImplementedError("MultiheadAttention with kdim / vdim different from embed_dim is not supported")
            cached = (AttentionProjection(module, 'attn_in'), AttentionProjection(module, 'attn_out'))
            module.__dict__["_curv_projections"] = cached
        return cached

    @property
    def weight(self):
        return self.module.in_proj_weight if self.kind == 'attn_in' else self.module.out_proj.weight

    @property
    def bias(self):
        return self.module.in_proj_bias if self.kind == 'attn_in' else self.module.out_proj.bias

    @property
    def _parameters(self):
        return {'weight': self.weight, 'bias': self.bias}

    @property
    def in_features(self) -> int:
        return self.weight.shape[1]

    @property
    def out_features(self) -> int:
        return self.weight.shape[0]

    def state_key(self, prefix: str, name: str) -> str:
        """Key of `name` ('weight' / 'bias') in the model's state_dict, `prefix` = qualified name of the module."""
        dot = prefix + "." if prefix else ""
        return dot + ("in_proj_" + name if self.kind == 'attn_in' else "out_proj." + name)

    def __repr__(self):
        return f"AttentionProjection({self.kind}, {self.in_features} -> {self.out_features})"


class _LinearTap:
    """While an ``nn.MultiheadAttention`` forward runs, the module-level name ``torch.nn.functional.linear`` is replaced
    by a wrapper that records, for the module's two projections, the input of the product and (through a tensor hook)
    the gradient of its output - what the forward / backward hooks of an ordinary Linear layer record
    (curvatures.py:306-310).  The attention forward calls F.linear directly (also for out_proj, whose own module hooks
    therefore never fire).  Installed by a forward pre-hook, removed by an always-called forward hook.  Several taps may
    be active at once (two estimators on one model, nested modules): ONE wrapper serves all of them and F.linear is
    restored when the last one leaves."""

    _active: List["_LinearTap"] = []
    _original = None

    def __init__(self, estimator, module: Module):
        self.estimator, self.module = estimator, module

    @staticmethod
    def _dispatch(input, weight, bias=None):
        out = _LinearTap._original(input, weight, bias)
        for tap in list(_LinearTap._active):
            proj_in, proj_out = AttentionProjection.of(tap.module)
            target = proj_in if weight is proj_in.weight else proj_out if weight is proj_out.weight else None
            if target is None and weight.data_ptr() == proj_in.weight.data_ptr() and weight.shape != proj_in.weight.shape:
                raise NotImplementedError("KFAC / EFB / INF support MultiheadAttention for self-attention only (query, key "
                                          "and value must be the same tensor)")
            if target is not None:
                record = tap.estimator.record
                record[target][0] = input
                if out.requires_grad:
                    out.register_hook(lambda grad, r=record, t=target: r[t].__setitem__(1, grad))
        return out

    def install(self, *_):
        import torch.nn.functional as F
        if self in _LinearTap._active:
            return
        if not _LinearTap._active:
            _LinearTap._original = F.linear
            F.linear = _LinearTap._dispatch
        _LinearTap._active.append(self)

    def remove(self, *_):
        import torch.nn.functional as F
        if self in _LinearTap._active:
            _LinearTap._active.remove(self)
            if not _LinearTap._active:
                F.linear = _LinearTap._original
                _LinearTap._original = None


class Curvature(ABC):
    """Base class: layer selection, mean weights, `_replace`, `sample_and_replace`.

    Mirrors curvature/curvatures.py:17-129.  Layers are selected by class NAME in ``model.modules()``
    order; that order is the layer index used by per-layer ``add`` / ``multiply`` lists."""

    # estimators whose reference implementation handles nn.MultiheadAttention (Diagonal only here; the
    # reference's KFAC / EFB raise NotImplementedError for it, curvatures.py:303-304, 435-436)
    _supports_mha = False
    # KFAC / EFB / INF: every selected MultiheadAttention module contributes its two projections as layers of their own
    # (`AttentionProjection`), an extension of the reference (which raises for them)
    _mha_as_projections = False

    def __init__(self, model: Union[Module, Sequential], layer_types: Union[List[str], str] = None, *,
                 shard=None):
        self.model = model
        self.model_state = copy.deepcopy(model.state_dict())
        self.layer_types = list()
        if isinstance(layer_types, str):
            self.layer_types.append(layer_types)
        elif isinstance(layer_types, list):
            self.layer_types.extend(layer_types if layer_types else SUPPORTED_LAYERS)
        elif layer_types is None:
            self.layer_types.extend(SUPPORTED_LAYERS)
        else:
            raise TypeError
        for _type in self.layer_types:
            assert _type in SUPPORTED_LAYERS
        self.state = dict()
        self.inv_state = dict()
        # optional layer sharding across ranks (curvature_amd.sharding.Shard); None = own every layer.
        # Keyword-only extension of the reference signature; may also be assigned after construction for
        # estimators whose constructor does no per-layer work (Diagonal, KFAC).
        self.shard = shard
        # device-side noise generator of the samplers (Philox4x32): `noise_offset` advances per draw.  The
        # seed is drawn from torch's default generator at the FIRST draw of this instance (the reference
        # draws its noise from that generator, so successive estimators and samples are independent and
        # torch.manual_seed() before sampling is honoured); assign `noise_seed` to pin it.
        self.noise_seed = None
        self.noise_offset = 0
        # the library's internal streams should exist before unrelated ones (RCCL's, a data loader's, eval_bnn's): the
        # estimator constructor is the earliest point at which the device is known (include/curv_hip.h: curv_init_streams)
        first = next(iter(model.parameters()), None)
        if first is not None and first.is_cuda:
            _lib.init_streams(first.device)

    # ------------------------------------------------------------------ helpers
    def _layers(self) -> List[Module]:
        """Selected Linear / Conv2d layers in ``model.modules()`` order (curvatures.py:120-122).  The walk over the
        module tree is done once per estimator (0.1 ms for a ResNet-50, paid by every phase of a step otherwise): the
        forward / backward hooks are registered on the layers found at construction, so layers added to the model
        later are outside the estimator here as they are in the reference."""
        cached = self.__dict__.get("_layers_cache")
        if cached is not None and cached[0] is self.model and cached[1] == tuple(self.layer_types):
            return list(cached[2])
        out = []
        for layer in self.model.modules():
            name = layer.__class__.__name__
            if name in self.layer_types:
                if name in ('Linear', 'Conv2d'):
                    out.append(layer)
                elif name == 'MultiheadAttention' and self._mha_as_projections:
                    out.extend(AttentionProjection.of(layer))
                elif name == 'MultiheadAttention' and not self._supports_mha:
                    raise NotImplementedError
        self.__dict__["_layers_cache"] = (self.model, tuple(self.layer_types), tuple(out))
        return out

    def _attention(self) -> List[Module]:
        """Selected MultiheadAttention modules in ``model.modules()`` order (Diagonal only)."""
        if 'MultiheadAttention' not in self.layer_types:
            return []
        return [l for l in self.model.modules() if l.__class__.__name__ == 'MultiheadAttention']

    def _owned(self):
        """[(global layer index, layer)] of the layers this rank owns (all of them without a shard)."""
        layers = self._layers()
        if self.shard is None:
            return list(enumerate(layers))
        return [(i, l) for i, l in enumerate(layers) if self.shard.owns(i)]

    def _global_index(self) -> Dict[Any, int]:
        """state key -> position in the reference's ``enumerate(self.state)`` order of an UNSHARDED run:
        first-seen order over ``model.modules()`` of the selected Linear / Conv2d layers (and, for Diagonal,
        the 'attn_in' / 'attn_out' keys).  This is the index of per-layer ``add`` / ``multiply`` lists
        (curvatures.py:184, 360, 444, 515); a rank that holds only its own layers in `state` still looks its
        hyper-parameters up by this global position."""
        order = []
        for layer in self.model.modules():
            name = layer.__class__.__name__
            if name not in self.layer_types:
                continue
            if name in ('Linear', 'Conv2d'):
                order.append(layer)
            elif name == 'MultiheadAttention' and self._mha_as_projections:
                order.extend(AttentionProjection.of(layer))
            elif name == 'MultiheadAttention' and self._supports_mha:
                order.extend(k for k in ('attn_in', 'attn_out') if k not in order)
        return {k: i for i, k in enumerate(order)}

    def _allgather_sampled(self):
        """Multi-GPU: the single collective of the path, reassembling every layer's sampled parameters.
        Attention modules (Diagonal only) are not part of the layer partition: rank 0 samples them and their
        projection parameters travel in the same all-gather, so that every rank ends with the same weights."""
        if self.shard is not None and (self.shard.world > 1 or self.shard.force_collective):
            entries = [[p.data for p in (l.weight, l.bias) if p is not None] for l in self._layers()]
            owners = list(self.shard.owner)
            for layer in (self._attention() if self._supports_mha else []):
                entries.append([p.data for p in (layer.in_proj_weight, layer.in_proj_bias, layer.out_proj.weight,
                                                 layer.out_proj.bias) if p is not None])
                owners.append(0)
            self.shard.allgather_params(entries, owners)

    @staticmethod
    def _hyper(add, multiply, index: int, count: int):
        """(n, s) of layer `index`: lists only when BOTH are non-scalars (curvatures.py:361-365)."""
        if not _is_scalar(add) and not _is_scalar(multiply):
            assert len(add) == len(multiply) == count
            return float(add[index]), float(multiply[index])
        return float(add), float(multiply)

    def _seed(self) -> int:
        if self.noise_seed is None:
            drawn = int(torch.empty((), dtype=torch.int64).random_().item())      # torch's default CPU generator
            rank = self.shard.rank if self.shard is not None else 0
            # ranks of a sharded run replay the same torch seeds (replicated forward/backward): decorrelate them
            self.noise_seed = (drawn + 0x9E3779B97F4A7C15 * rank) & (2 ** 63 - 1)
        return self.noise_seed

    def _randn(self, *shape, device, out: Optional[Tensor] = None) -> Tensor:
        numel = 1
        for s in shape:
            numel *= int(s)
        counter = getattr(self, "_noise_counter", None)
        if counter is not None:
            # stream position on the device (curvature_amd.graph): the same draws as with the host-side offset, but a
            # captured replay advances it too
            return ops.randn(shape, device, self._seed(), out=out, counter=counter)
        out = ops.randn(shape, device, self._seed(), self.noise_offset, out=out)
        self.noise_offset += (numel + 3) // 4
        return out

    def use_device_noise_counter(self, enable: bool = True) -> None:
        """Keep the position of the noise stream in a device word instead of `noise_offset` (needed inside a captured
        HIP graph, where a host-side offset would be frozen).  Switching back reads the word once (a host sync)."""
        counter = getattr(self, "_noise_counter", None)
        if enable and counter is None:
            dev = next(self.model.parameters()).device
            self._seed()
            self._noise_counter = torch.tensor([self.noise_offset], dtype=torch.int64, device=dev)
        elif not enable and counter is not None:
            self.noise_offset = int(counter.item())
            self._noise_counter = None

    def _sample_plans(self) -> dict:
        """Launch plans of sample_and_replace, keyed by the addresses they were described for.  TWO are kept: the
        overlapped inference loop (evaluate.eval_bnn) alternates between two parameter buffer sets."""
        return self.__dict__.setdefault("_sample_plan_cache", {})

    def _keep_plan(self, key, plan) -> None:
        plans = self._sample_plans()
        while len(plans) >= 2:
            plans.pop(next(iter(plans)))
        plans[key] = plan

    def _reload_mean(self, skip: Sequence[Tensor] = ()):
        """``model.load_state_dict(model_state)`` (curvatures.py:119) as one batched copy: a ResNet-50 has
        ~320 state tensors, i.e. ~320 copy launches (3 ms) through torch.  `skip`: live tensors the
        caller overwrites completely right afterwards."""
        live = getattr(self, "_reload_live", None)
        if live is None:
            state = self.model.state_dict(keep_vars=True)
            if list(state.keys()) != list(self.model_state.keys()):
                raise RuntimeError("model structure changed since the estimator was created")
            self._reload_live = live = [(k, v) for k, v in state.items()]
            self._reload_keys = [k for k, _ in live]
            self._reload_tensors = [v for _, v in live]
            self._reload_plans = {}
        if not live or not live[0][1].is_cuda:
            self.model.load_state_dict(self.model_state)     # CPU models: torch plumbing, nothing to batch
            return
        # parameters may have been re-homed (.to(), ...) and model_state may have been reassigned: the plan is keyed on
        # every address (C-level loops: this runs in front of every sample of a BNN loop, with the GPU waiting)
        ptr = Tensor.data_ptr
        ms = self.model_state
        ptrs = tuple(map(ptr, self._reload_tensors))
        means = tuple(map(ptr, [ms[k] for k in self._reload_keys]))
        key = (ptrs, means, tuple(sorted(map(ptr, skip))))
        plan = self._reload_plans.get(key)
        if plan is None:
            while len(self._reload_plans) >= 2:                  # two parameter buffer sets (evaluate.eval_bnn)
                self._reload_plans.pop(next(iter(self._reload_plans)))
            skipped = set(key[2])
            pairs = [(v.data, self.model_state[k]) for k, v in live if v.data_ptr() not in skipped]
            plan = ops.CopyPlan([d for d, _ in pairs], [s_ for _, s_ in pairs])
            self._reload_plans[key] = plan
        plan.run()

    def model_state_of(self, layer: Module, name: str) -> Tensor:
        """The mean (MAP) tensor of `layer.<name>` inside ``model_state``."""
        if not hasattr(self, "_state_keys"):
            self._state_keys = {}
            for prefix, mod in self.model.named_modules():
                for pname, _ in mod.named_parameters(recurse=False):
                    self._state_keys[(mod, pname)] = (prefix + "." if prefix else "") + pname
                if mod.__class__.__name__ == 'MultiheadAttention' and "_curv_projections" in mod.__dict__:
                    for proj in mod.__dict__["_curv_projections"]:
                        for pname in ('weight', 'bias'):
                            self._state_keys[(proj, pname)] = proj.state_key(prefix, pname)
        return self.model_state[self._state_keys[(layer, name)]]

    @staticmethod
    def _replace(sample: Tensor, weight: Tensor, bias: Tensor = None):
        """weight += sample[:, :-1], bias += sample[:, -1] (curvatures.py:67-82)."""
        if bias is not None:
            bias_sample = sample[:, -1].contiguous().view(*bias.shape)
            bias.data.add_(bias_sample)
            sample = sample[:, :-1]
        weight.data.add_(sample.contiguous().view(*weight.shape))

    # ------------------------------------------------------------------ plugin API
    @abstractmethod
    def update(self, *args: Any, **kwargs: Any):
        raise NotImplementedError

    @abstractmethod
    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        raise NotImplementedError

    @abstractmethod
    def sample(self, layer: Module) -> Tensor:
        raise NotImplementedError

    def sample_many(self, count: int) -> "SampleBank":
        """`count` sampled parameter sets of the estimator's own layers as a `SampleBank` (see `KFAC.sample_many`, which
        produces them in two launches; this generic form runs `sample_and_replace` `count` times and files the
        parameters away, so that `evaluate.eval_bnn(samples_per_launch=...)` works with every estimator)."""
        S = int(count)
        assert S >= 1
        owned = [l for _, l in self._owned() if getattr(l, "weight", None) is not None]
        # every parameter sample_and_replace modifies goes into the bank: Diagonal also samples the projections of the
        # selected nn.MultiheadAttention modules (string keys, outside `_layers()`); left out, `replace_from` would reset
        # them to their means
        for mha in (self._attention() if self._supports_mha else []):
            owned += [_InProjection.of(mha), mha.out_proj]
        weights = {l: torch.empty(S, *l.weight.shape, dtype=l.weight.dtype, device=l.weight.device) for l in owned}
        biases = {l: (torch.empty(S, *l.bias.shape, dtype=l.bias.dtype, device=l.bias.device) if l.bias is not None else None)
                  for l in owned}
        gather, self._allgather_sampled = self._allgather_sampled, (lambda: None)     # the sets travel when they are loaded
        try:
            for k in range(S):
                self.sample_and_replace()
                for l in owned:
                    weights[l][k].copy_(l.weight.data)
                    if biases[l] is not None:
                        biases[l][k].copy_(l.bias.data)
        finally:
            self._allgather_sampled = gather
        return SampleBank(S, weights, biases)

    def replace_from(self, bank: "SampleBank", index: int) -> None:
        """Load parameter set `index` of a `sample_many` bank into the model (the other state tensors go back to their
        means; under a layer shard the sets of all ranks are all-gathered as in `sample_and_replace`)."""
        if not 0 <= index < bank.count:
            raise IndexError("replace_from: sample index out of range")
        plans = bank.__dict__.setdefault("_copy_plans", {})
        layers = list(bank.weights.keys())
        params = [p for l in layers for p in (l._parameters['weight'], l._parameters['bias']) if p is not None]
        key = (index, tuple(map(Tensor.data_ptr, params)))
        plan = plans.get(key)
        if plan is None:
            dsts, srcs = [], []
            for layer in layers:
                dsts.append(layer.weight.data)
                srcs.append(bank.weights[layer][index].view(layer.weight.shape))
                if layer.bias is not None:
                    dsts.append(layer.bias.data)
                    srcs.append(bank.biases[layer][index])
            plan = plans[key] = ops.CopyPlan(dsts, srcs)
        plan.run()
        self._reload_mean(skip=params)
        self._allgather_sampled()

    def sample_and_replace(self):
        """Reset to the mean weights, then add one posterior sample per selected layer (curvatures.py:117-129)."""
        self._reload_mean()
        for _, layer in self._owned():
            _sample = self.sample(layer)
            self._replace(_sample, layer.weight, layer.bias)
        self._allgather_sampled()


class Diagonal(Curvature):
    """Diagonal Fisher: state += grad**2 * batch_size (curvatures.py:132-193).

    ``nn.MultiheadAttention`` modules are handled as in the reference (:159-174, 125-129): their input and
    output projections accumulate under the string keys ``'attn_in'`` / ``'attn_out'`` (ONE pair of keys for
    the whole model, as in the reference).  With a layer shard every rank owns a disjoint set of the
    Linear / Conv2d layers; the attention entries are small: every rank accumulates and inverts them, rank 0
    draws their sample and the all-gather of `sample_and_replace` distributes it (each rank has its own noise
    stream, so sampling them everywhere would leave the ranks with different attention weights)."""

    _supports_mha = True

    def update(self, batch_size: int):
        # one pass over modules() so that `state` gets the reference's insertion order (curvatures.py:149-174),
        # which is the order per-layer hyper-parameter lists are indexed by
        owned = {l for _, l in self._owned()}
        # the state of all (new) Linear / Conv2d layers in one arena: invert() with one pair of hyper-parameters is
        # then a single launch over it
        new = [l for _, l in self._owned() if l not in self.state and l.weight.grad is not None]
        fresh = {}
        if new:
            self._state_flat, views = _arena(
                [(l.weight.shape[0], l.weight.numel() // l.weight.shape[0] + int(l.bias is not None)) for l in new],
                new[0].weight.device)
            fresh = dict(zip(new, views))
        keys, items = [], []
        for layer in self.model.modules():
            name = layer.__class__.__name__
            if name not in self.layer_types:
                continue
            if name in ('Linear', 'Conv2d'):
                if layer in owned:
                    bias_grad = layer.bias.grad.contiguous() if layer.bias is not None else None
                    keys.append(layer)
                    items.append((layer.weight.grad.contiguous(), bias_grad, fresh.get(layer, self.state.get(layer)),
                                  True if layer in fresh else None))
            elif name == 'MultiheadAttention':
                for key, weight, bias in (('attn_in', layer.in_proj_weight, layer.in_proj_bias),
                                          ('attn_out', layer.out_proj.weight, layer.out_proj.bias)):
                    keys.append(key)
                    items.append((weight.grad.contiguous(), bias.grad.contiguous(), self.state.get(key), None))
        # several modules may share the 'attn_*' keys (one pair for the whole model, as in the reference): the first
        # occurrence of every key goes into ONE launch, later ones accumulate onto it afterwards
        firsts, later = {}, []
        for key, item in zip(keys, items):
            if key in firsts:
                later.append((key, item))
            else:
                firsts[key] = item
        for key, st in zip(firsts, ops.sq_accumulate_many(firsts.values(), batch_size)):
            self.state[key] = st                   # new keys enter in modules() order, like the reference's
        for key, item in later:
            self.state[key] = ops.sq_accumulate(item[0], item[1], batch_size, self.state[key])

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        gindex = self._global_index()
        # first inversion: all inverse-state tensors of the Linear / Conv2d layers in ONE arena, so that sampling can
        # scale the whole model's noise with one launch (later calls overwrite them in place)
        arena = {}
        fresh = [l for l in self.state if l not in self.inv_state and not isinstance(l, str)]
        if fresh:
            self._inv_flat, views = _arena([tuple(self.state[l].shape) for l in fresh], self.state[fresh[0]].device)
            arena = dict(zip(fresh, views))
        plain = [l for l in self.state if not isinstance(l, str)]
        state_flat, inv_flat = getattr(self, "_state_flat", None), getattr(self, "_inv_flat", None)
        whole = _is_scalar(add) and _is_scalar(multiply) and len(plain) == len(self.state) and \
            _is_arena(state_flat, [self.state[l] for l in plain]) and \
            _is_arena(inv_flat, [self.inv_state.get(l, arena.get(l)) for l in plain]) and \
            state_flat.numel() == inv_flat.numel() == sum(self.state[l].numel() for l in plain)
        if whole:      # one pair of hyper-parameters, both dicts whole arenas (no attention entries): one launch
            for l in plain:
                self.inv_state.setdefault(l, arena.get(l))
            ops.rsqrt_affine(state_flat, float(add), float(multiply), out=inv_flat)
            return
        for position, (layer, value) in enumerate(self.state.items()):
            # Diagonal uses lists when both are list/tuple (curvatures.py:183); same outcome as _hyper.
            # Keys that are not layers of this model (a foreign state dict) fall back to their position.
            n, s = self._hyper(add, multiply, gindex.get(layer, position), max(len(gindex), len(self.state)))
            out = self._reuse(self.inv_state.get(layer), value)
            if out is None and layer in arena:
                out = arena[layer]
            self.inv_state[layer] = ops.rsqrt_affine(value, n, s, out=out)

    @staticmethod
    def _reuse(prev: Optional[Tensor], like: Tensor) -> Optional[Tensor]:
        """The previous inverse-state tensor if it can be overwritten in place (stable addresses keep cached
        launch plans valid)."""
        if prev is not None and prev.shape == like.shape and prev.device == like.device and prev.is_contiguous():
            return prev
        return None

    def sample(self, layer: Union[Module, str], z: Optional[Tensor] = None) -> Tensor:
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        inv = self.inv_state[layer]
        if z is None:
            z = self._randn(*inv.shape, device=inv.device)
        return ops.mul(z, inv)

    def sample_and_replace(self):
        """curvatures.py:117-129 including the MultiheadAttention branch.  For the Linear / Conv2d layers the
        per-layer loop (draw, multiply, two adds: ~5 launches per layer, launch-bound for a ResNet) is one noise
        launch, one multiply over an arena and one batched launch that writes ``mean + z * inv_state`` through the
        [W | b] split straight onto the parameters (a K = 1 product with the `MUL_E_ADD_F` epilogue)."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        layers = [l for _, l in self._owned() if l in self.inv_state]
        self._reload_mean(skip=[p for l in layers for p in (l.weight, l.bias) if p is not None])
        if layers:
            key = (tuple(self.inv_state[l].data_ptr() for l in layers),
                   tuple(p.data_ptr() for l in layers for p in (l.weight, l.bias) if p is not None),
                   tuple(self.model_state_of(l, nm).data_ptr() for l in layers for nm in ('weight', 'bias')
                         if getattr(l, nm) is not None))
            plan = self._sample_plans().get(key)
            if plan is None:
                dev = self.inv_state[layers[0]].device
                zflat, zs = _arena([tuple(self.inv_state[l].shape) for l in layers], dev)
                rows = max(self.inv_state[l].shape[0] for l in layers)
                cols = max(self.inv_state[l].shape[1] for l in layers)
                ones_r = torch.ones(rows, 1, dtype=torch.float32, device=dev)
                ones_c = torch.ones(1, cols, dtype=torch.float32, device=dev)
                scale, jobs = [], []
                for layer, z in zip(layers, zs):
                    m, n = z.shape
                    n0 = n - int(layer.bias is not None)
                    scale.append((z, self.inv_state[layer]))
                    w = layer.weight.data
                    if not w.is_contiguous():
                        raise RuntimeError("Diagonal.sample_and_replace: parameters must be contiguous")
                    jobs.append(ops.Gemm(ones_r[:m], ones_c[:, :n0], w.view(m, n0), epilogue=ops.EPI_MUL_E_ADD_F,
                                         E=z[:, :n0], F=self.model_state_of(layer, 'weight').view(m, n0)))
                    if layer.bias is not None:
                        jobs.append(ops.Gemm(ones_r[:m], ones_c[:, :1], layer.bias.data.view(m, 1),
                                             epilogue=ops.EPI_MUL_E_ADD_F, E=z[:, n0:],
                                             F=self.model_state_of(layer, 'bias').view(m, 1)))
                inv_flat = getattr(self, "_inv_flat", None)
                whole = _is_arena(inv_flat, [self.inv_state[l] for l in layers]) and \
                    sum(self.inv_state[l].numel() for l in layers) == zflat.numel()
                plan = (key, zflat, scale, ops.GemmPlan(jobs), inv_flat if whole else None)
                self._keep_plan(key, plan)
            _, zflat, scale, gemms, inv_flat = plan
            self._randn(zflat.numel(), device=zflat.device, out=zflat)
            if inv_flat is not None:                               # z *= inv_state: one launch over the arena
                ops.mul(zflat, inv_flat[:zflat.numel()], out=zflat)
            else:
                for z, inv in scale:
                    ops.mul(z, inv, out=z)
            gemms.run()
        if self.shard is None or self.shard.rank == 0:         # one owner for the attention entries
            for layer in self._attention():
                for weight, bias, key in ((layer.in_proj_weight, layer.in_proj_bias, 'attn_in'),
                                          (layer.out_proj.weight, layer.out_proj.bias, 'attn_out')):
                    self._replace(self.sample(key), weight, bias)
        self._allgather_sampled()


class BlockDiagonal(Curvature):
    """Block-diagonal (per-layer, full P x P) Fisher: state += ger(g, g) * batch_size with
    g = [W.grad.view(-1) ; b.grad] (curvatures.py:196-261).

    * ``update``: the rank-1 accumulation is the factor-build kernel on a one-sample linear "layer" (src (1, P)).
    * ``invert``: ``(s F + n I).inverse().cholesky()`` (:252-253) is the KFAC factor inversion with (n^2, s^2)
      passed for (add, multiply) - that routine damps with sqrt(s) F + sqrt(n) I, computed in double on the host.
    * ``sample``: ``z @ L`` (:258) as one GEMM.  The reference then views the weight part with ``weight.shape`` and
      concatenates the bias column along dim 1, which raises for Conv2d (4-D with 2-D).  Here the weight part is
      (out, -1) for every layer type - identical for Linear, and what ``_replace`` consumes for Conv2d.
    O(P^2) memory per layer: meant for small layers, like the reference's.  MultiheadAttention is not supported
    (the reference's branch, :220-239, concatenates a 2-D gradient with a 1-D bias and raises)."""

    def update(self, batch_size: int):
        jobs = []
        for _, layer in self._owned():
            parts = [layer.weight.grad.contiguous().view(-1)]
            if layer.bias is not None:
                parts.append(layer.bias.grad.contiguous())
            g = ops.concat(parts) if len(parts) > 1 else parts[0]
            first = layer not in self.state
            if first:
                self.state[layer] = torch.empty(g.numel(), g.numel(), dtype=torch.float32, device=g.device)
            jobs.append(ops.FactorJob(g.view(1, -1), self.state[layer], scale=float(batch_size), first=first))
        ops.kfac_accumulate(jobs)

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        gindex = self._global_index()
        adds, muls = [], []
        for position, layer in enumerate(self.state.keys()):
            n, s = self._hyper(add, multiply, gindex.get(layer, position), max(len(gindex), len(self.state)))
            adds.append(n * n)
            muls.append(s * s)
        prev = [self.inv_state.get(layer) for layer in self.state.keys()]
        chols = ops.chol_inv_lower(list(self.state.values()), adds, muls, outs=prev)
        for layer, chol in zip(self.state.keys(), chols):
            self.inv_state[layer] = chol

    def _draw(self, layer: Module, z: Optional[Tensor]) -> Tensor:
        inv = self.inv_state[layer]
        if z is None:
            z = self._randn(inv.shape[0], device=inv.device)
        x = torch.empty(1, inv.shape[0], dtype=torch.float32, device=inv.device)
        ops.gemm_batched([ops.Gemm(z.view(1, -1), inv, x)])
        return x.view(-1)

    def sample(self, layer: Module, z: Optional[Tensor] = None) -> Tensor:
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        x = self._draw(layer, z)
        n_w = layer.weight.numel()
        rows = layer.weight.shape[0]
        out = torch.empty(rows, n_w // rows + int(layer.bias is not None), dtype=torch.float32, device=x.device)
        out[:, :n_w // rows] = x[:n_w].view(rows, -1)         # the reference's torch.cat (:260-261): API plumbing
        if layer.bias is not None:
            out[:, -1] = x[n_w:]
        return out

    def sample_and_replace(self):
        """The base-class loop fused: one noise launch and one batched GEMM launch for all owned layers.  The draw
        z @ L is written straight onto the parameters in g's order (weights, then biases) with the mean added in the
        epilogue: no [W | b] detour, and the reload of the mean skips the parameters that are overwritten here."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        owned = [(i, l) for i, l in self._owned() if l in self.inv_state]
        jobs, skip = [], []
        if owned:
            dev = self.inv_state[owned[0][1]].device
            flat = self._randn(sum(self.inv_state[l].shape[0] for _, l in owned), device=dev)
            pos = 0
            for _, layer in owned:
                inv = self.inv_state[layer]
                P, n_w = inv.shape[0], layer.weight.numel()
                z = flat[pos:pos + P].view(1, P)
                pos += P
                w = layer.weight.data
                if not w.is_contiguous():
                    raise RuntimeError("BlockDiagonal.sample_and_replace: parameters must be contiguous")
                jobs.append(ops.Gemm(z, inv[:, :n_w], w.view(1, n_w), epilogue=ops.EPI_ADD_E,
                                     E=self.model_state_of(layer, 'weight').view(1, n_w)))
                skip.append(w)
                if layer.bias is not None:
                    b = layer.bias.data
                    jobs.append(ops.Gemm(z, inv[:, n_w:], b.view(1, P - n_w), epilogue=ops.EPI_ADD_E,
                                         E=self.model_state_of(layer, 'bias').view(1, P - n_w)))
                    skip.append(b)
        self._reload_mean(skip=skip)
        ops.gemm_batched(jobs)
        self._allgather_sampled()


class KFAC(Curvature):
    """Kronecker-factored Fisher (curvatures.py:264-392).

    ``state[layer] = [A, G]`` (fp32, exactly symmetric), ``inv_state[layer] = (L_A, L_G)`` with
    L = chol_lower((sqrt(s) F + sqrt(n) I)^-1).  ``record[layer] = [input, grad_output]``: unlike the
    reference the recorded grad_output is NOT pre-multiplied by the batch size (curvatures.py:310); the
    factor N is folded into the scale of the G-side SYRK, which saves one pass over every gradient."""

    _mha_as_projections = True

    def __init__(self, model: Union[Module, Sequential], layer_types: Union[List[str], str] = None, *, shard=None):
        super().__init__(model, layer_types, shard=shard)
        self.hooks = list()
        self.record = dict()
        for layer in model.modules():
            name = layer.__class__.__name__
            if name in self.layer_types:
                if name in ('Linear', 'Conv2d'):
                    if name == 'Conv2d' and (tuple(layer.dilation) != (1, 1) or layer.groups != 1):
                        # the reference silently ignores both (curvatures.py:329)
                        raise NotImplementedError("KFAC: dilated or grouped convolutions are not supported")
                    if name == 'Conv2d' and not all(isinstance(p, int) for p in layer.padding):
                        raise NotImplementedError("KFAC: string padding modes are not supported")
                    self.record[layer] = [None, None]
                    self.hooks.append(layer.register_forward_pre_hook(self._save_input))
                    self.hooks.append(layer.register_forward_hook(self._hook_output))
                elif name == 'MultiheadAttention':
                    # the two projections as Linear-like layers (extension: curvatures.py:303-304 raises here); their
                    # inputs / output gradients are tapped off the F.linear calls of the attention forward
                    tap = _LinearTap(self, layer)
                    for proj in AttentionProjection.of(layer):
                        self.record[proj] = [None, None]
                    self.hooks.append(layer.register_forward_pre_hook(tap.install))
                    self.hooks.append(layer.register_forward_hook(tap.remove, always_call=True))

    def _save_input(self, module, input):
        self.record[module][0] = input[0]            # by reference, like curvatures.py:307

    def _hook_output(self, module, input, output):
        if output.requires_grad:
            output.register_hook(lambda grad, module=module: self._save_output(module, grad))

    def _save_output(self, module, grad_output):
        self.record[module][1] = grad_output         # raw; the reference stores grad * N (curvatures.py:310)

    def update(self, batch_size: int = None, *, inputs: bool = True, grads: bool = True, input_weight: float = 1.0):
        """A += X X^T / (N L), G += (N g)(N g)^T / (N L) for every selected layer: one grouped launch.

        The keyword-only arguments extend the reference's ``update(batch_size)`` (curvatures.py:312) for
        Monte-Carlo Fisher loops that run several backward passes per forward pass
        (``curvature_amd.factors.compute_factors``): the A side depends only on the layer inputs, so it is
        built once per forward with ``input_weight`` = number of backward passes (``inputs=False`` for the
        others), instead of adding the same matrix again and again."""
        jobs = []
        fresh = getattr(self, "_fresh", None)
        if fresh is None:
            fresh = self._fresh = set()              # factors allocated here that nothing has written yet
        for _, layer in self._owned():
            forward, backward = self.record[layer]
            if (inputs and forward is None) or (grads and backward is None):
                raise RuntimeError("KFAC.update: no recorded forward/backward pass for a selected layer")
            has_bias = layer.bias is not None
            x = g = None
            if forward is not None:
                x = forward.detach()
                if x.dtype != torch.float32:
                    raise RuntimeError("KFAC.update expects float32 activations and gradients")
                x = x.contiguous()
            if backward is not None:
                g = backward.detach()
                if g.dtype != torch.float32:
                    raise RuntimeError("KFAC.update expects float32 activations and gradients")
                g = g.contiguous()
            if layer.__class__.__name__ == 'Conv2d':
                kernel, stride, padding = layer.kernel_size, layer.stride, layer.padding
                C, m = layer.in_channels, layer.out_channels
                N = (x if x is not None else g).shape[0]
                if g is not None:
                    L = g.shape[2] * g.shape[3]
                else:
                    L = ((x.shape[2] + 2 * padding[0] - kernel[0]) // stride[0] + 1) * \
                        ((x.shape[3] + 2 * padding[1] - kernel[1]) // stride[1] + 1)
                n = C * kernel[0] * kernel[1] + int(has_bias)
            else:
                if x is not None and x.dim() != 2:      # (N, *, in) inputs: flatten the leading dims
                    x = x.reshape(-1, x.shape[-1])
                if g is not None and g.dim() != 2:
                    g = g.reshape(-1, g.shape[-1])
                N = (x if x is not None else g).shape[0]
                C, m = layer.in_features, layer.out_features
                kernel, stride, padding, L = (1, 1), (1, 1), (0, 0), 1
                n = C + int(has_bias)
            dev = (x if x is not None else g).device
            if layer not in self.state:
                # a side that is not written by this call must start from zero, not from garbage
                alloc = torch.empty if (inputs and grads) else torch.zeros
                self.state[layer] = [alloc(n, n, dtype=torch.float32, device=dev),
                                     alloc(m, m, dtype=torch.float32, device=dev)]
                if inputs and grads:
                    fresh.update(((layer, 0), (layer, 1)))
            A, G = self.state[layer]
            if inputs:
                first = (layer, 0) in fresh
                fresh.discard((layer, 0))
                jobs.append(ops.FactorJob(x, A, kernel, stride, padding, has_bias, float(input_weight) / (N * L), first))
            if grads:
                first = (layer, 1) in fresh
                fresh.discard((layer, 1))
                jobs.append(ops.FactorJob(g, G, (1, 1), (1, 1), (0, 0), False, float(N) / L, first))
        if self.shard is not None and self.shard.world > 1:
            # the launch form is a property of the MODEL, not of this rank's share: a share under the small-launch threshold
            # would otherwise sum its factors in another order than the unsharded run (which is over it)
            # (decided by the library itself - curv_kfac_path_for evaluates every gate of the small form, not only the
            # flops - on the geometry of ALL selected layers: the hooks record every layer on every rank)
            geoms, known = [], True
            for layer in self._layers():
                forward, backward = self.record[layer]
                if (inputs and forward is None) or (grads and backward is None):
                    known = False
                    break
                if layer.__class__.__name__ == 'Conv2d':
                    if inputs:
                        geoms.append((*forward.shape, layer.kernel_size, layer.stride, layer.padding, layer.bias is not None))
                    if grads:
                        geoms.append((*backward.shape, (1, 1), (1, 1), (0, 0), False))
                else:
                    if inputs:
                        geoms.append((forward.numel() // forward.shape[-1], forward.shape[-1], 1, 1, (1, 1), (1, 1), (0, 0),
                                      layer.bias is not None))
                    if grads:
                        geoms.append((backward.numel() // backward.shape[-1], backward.shape[-1], 1, 1, (1, 1), (1, 1), (0, 0), False))
            hint = ops.kfac_path_for(geoms) if known else _lib.PATH_GROUPED
            for job in jobs:
                job.path_hint = hint
        if getattr(self, "_count_flops", False):                 # bench.py: what the launch plan executes
            self._last_flops = sum(ops.kfac_plan_flops(jobs))
        ops.kfac_accumulate(jobs, events=getattr(self, "_timing_events", None))

    def restart_accumulation(self) -> None:
        """The next `update()` overwrites the factors of every layer instead of adding to them (the tensors, their
        addresses and the launch plans built on them stay).  Extension of the reference API: its only way to start
        over is a new estimator."""
        fresh = getattr(self, "_fresh", None)
        if fresh is None:
            fresh = self._fresh = set()
        fresh.update((layer, side) for layer in self.state for side in (0, 1))

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1., *, check: bool = True):
        """`check=False` (keyword-only extension): skip the read-back of the status words - the call's only host
        synchronisation - and leave them for `check_invert()`; a HIP-graph capture of the step needs that."""
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        factors, adds, muls = [], [], []
        gindex = self._global_index()
        for position, (layer, value) in enumerate(self.state.items()):
            # layer index = position among the selected layers in modules() order (== enumerate(state)
            # of the reference, curvatures.py:360, when every layer is owned)
            n, s = self._hyper(add, multiply, gindex.get(layer, position), max(len(gindex), len(self.state)))
            for factor in value:
                factors.append(factor)
                adds.append(n)
                muls.append(s)
        # outputs of the previous call are overwritten in place (stable addresses keep the cached launch
        # plan of sample_and_replace valid); RuntimeError if a damped factor is not positive definite
        prev = [t for layer in self.state.keys() for t in self.inv_state.get(layer, (None, None))]
        chols = ops.chol_inv_lower(factors, adds, muls, check=check, outs=prev)
        self._invert_info = chols.info
        for index, layer in enumerate(self.state.keys()):
            self.inv_state[layer] = (chols[2 * index], chols[2 * index + 1])

    def check_invert(self) -> None:
        """Raise ``RuntimeError`` if the last ``invert(check=False)`` (or the last replay of a graph that contains it)
        met a damped factor that is not positive definite (curvatures.py:377-383 raises at that point)."""
        info = getattr(self, "_invert_info", None)
        if info is not None:
            ops.check_chol_info(info)

    def sample(self, layer: Module, z: Optional[Tensor] = None) -> Tensor:
        """(L_A z L_G^T)^T -> (m, n) (curvatures.py:387-392); `z` (n, m) may be supplied for parity tests."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        first, second = self.inv_state[layer]
        n, m = first.size(0), second.size(0)
        if z is None:
            z = self._randn(n, m, device=first.device)
        tmp = torch.empty(m, n, dtype=torch.float32, device=first.device)
        out = torch.empty(m, n, dtype=torch.float32, device=first.device)
        ops.gemm_batched([ops.Gemm(second, z.t(), tmp, tri=ops.TRI_A_LOWER)])       # L_G lower triangular
        ops.gemm_batched([ops.Gemm(tmp, first.t(), out, tri=ops.TRI_B_UPPER)])      # L_A^T upper triangular
        return out

    def sample_and_replace(self, noise: Optional[Dict[Module, Tensor]] = None):
        """Fused form of the base-class loop: two batched GEMM launches for the whole model, the second
        writing ``mean + sample`` straight into the parameters (same result as curvatures.py:117-129)."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        owned = self._owned()
        ptr = Tensor.data_ptr
        params = [p for _, l in owned for p in (l._parameters['weight'], l._parameters['bias']) if p is not None]
        # The two GEMM launches are described once and replayed while the tensors involved stay where they
        # are (invert() rewrites inv_state in place): per call only the noise is drawn.
        inv_state = self.inv_state
        means = [self.model_state_of(l, nm) for _, l in owned for nm in ('weight', 'bias') if l._parameters[nm] is not None]
        key = (noise is None, tuple(map(ptr, [t for _, l in owned for t in inv_state[l]])), tuple(map(ptr, params)),
               tuple(map(ptr, means)), tuple(map(ptr, noise.values())) if noise is not None else ())
        plan = self._sample_plans().get(key)
        if plan is None:
            stage1, stage2 = [], []
            flat, pos = None, 0
            if noise is None and owned:        # one generator launch for the whole model
                dev = self.inv_state[owned[0][1]][0].device
                total = sum(self.inv_state[l][0].size(0) * self.inv_state[l][1].size(0) for _, l in owned)
                flat = torch.empty(total, dtype=torch.float32, device=dev)
            for _, layer in owned:
                first, second = self.inv_state[layer]
                n, m = first.size(0), second.size(0)
                if noise is not None:
                    z = noise[layer]
                else:
                    z = flat[pos:pos + n * m].view(n, m)
                    pos += n * m
                tmp = torch.empty(m, n, dtype=torch.float32, device=first.device)
                stage1.append(ops.Gemm(second, z.t(), tmp, tri=ops.TRI_A_LOWER))
                n0 = n - int(layer.bias is not None)
                w = layer.weight.data.view(m, n0)
                w_mean = self.model_state_of(layer, 'weight').view(m, n0)
                la_t = first.t()
                stage2.append(ops.Gemm(tmp, la_t[:, :n0], w, epilogue=ops.EPI_ADD_E, E=w_mean, tri=ops.TRI_B_UPPER))
                if layer.bias is not None:
                    b = layer.bias.data.view(m, 1)
                    b_mean = self.model_state_of(layer, 'bias').view(m, 1)
                    stage2.append(ops.Gemm(tmp, la_t[:, n0:], b, epilogue=ops.EPI_ADD_E, E=b_mean))
            # largest products first: the tail of each launch is then made of the short tiles
            stage1.sort(key=lambda j: -(j.A.shape[0] * j.A.shape[1] * j.B.shape[1]))
            stage2.sort(key=lambda j: -(j.A.shape[0] * j.A.shape[1] * j.B.shape[1]))
            plan = (key, flat, ops.GemmPlan(stage1), ops.GemmPlan(stage2))
            self._keep_plan(key, plan)
        if plan[1] is not None:
            self._randn(plan[1].numel(), device=plan[1].device, out=plan[1])
        plan[2].run()
        plan[3].run()
        # the other state tensors (BatchNorm, layers outside the estimator) go back to their means BEHIND the GEMMs: the
        # second stage has written weight and bias of every owned layer (mean + sample), the copies touch the rest, and
        # a step that has just synchronised in invert() gets its long kernels queued ~0.1 ms earlier this way
        self._reload_mean(skip=params)
        self._allgather_sampled()

    # ------------------------------------------------------------------ batched multi-sample generation
    def sample_many(self, count: int, noise: Optional[Dict[Module, Tensor]] = None) -> "SampleBank":
        """`count` posterior samples of every owned layer in two launches (scripts/evaluate.py:134-139 draws one sample
        per forward sweep; a BNN evaluation needs 10-100 of them).

        With z_s (n x m) the noise of sample s, W_s = (L_A z_s L_G^T)^T = L_G (L_A z_s)^T (curvatures.py:387-392).
        Stage A forms V = L_A [z_1 | ... | z_S] as ONE product per layer with S m columns - the n x n triangular factor
        (80 % of a ResNet-50 sample's flops) is streamed once for all S samples instead of once per sample - and stage
        B the S products W_s = L_G V_s^T with ``mean +`` fused, written into a bank of parameter sets
        (`SampleBank.weights[layer]`: (S, m, n0), `.biases[layer]`: (S, m)).  `replace_from(bank, s)` loads set s into
        the model.  `noise[layer]`: (S, n, m) caller-supplied z_s (parity tests); otherwise one generator launch."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        S = int(count)
        assert S >= 1
        owned = self._owned()
        ptr = Tensor.data_ptr
        key = ("many", S, noise is None, tuple(map(ptr, [t for _, l in owned for t in self.inv_state[l]])),
               tuple(map(ptr, noise.values())) if noise is not None else ())
        cache = self.__dict__.setdefault("_many_plans", {})
        plan = cache.get(key)
        if plan is None:
            cache.clear()                                    # one bank's worth of buffers at a time
            dev = self.inv_state[owned[0][1]][0].device
            total = sum(self.inv_state[l][0].size(0) * self.inv_state[l][1].size(0) for _, l in owned)
            flat = torch.empty(S * total, dtype=torch.float32, device=dev) if noise is None else None
            stage_a, stage_b, weights, biases, keep = [], [], {}, {}, []
            pos = 0
            for _, layer in owned:
                first, second = self.inv_state[layer]
                n, m = first.size(0), second.size(0)
                if noise is not None:
                    z = noise[layer]
                    if tuple(z.shape) != (S, n, m):
                        raise RuntimeError(f"sample_many: noise of a layer must be ({S}, {n}, {m})")
                    zt = z.transpose(1, 2).contiguous().view(S * m, n)          # rows (s, j): z_s^T
                else:
                    zt = flat[pos:pos + S * n * m].view(S * m, n)                 # iid: drawn directly as z_s^T
                    pos += S * n * m
                V = torch.empty(n, S * m, dtype=torch.float32, device=dev)
                stage_a.append(ops.Gemm(first, zt.t(), V, tri=ops.TRI_A_LOWER))     # V = L_A [z_1 | ... | z_S]
                has_bias = layer.bias is not None
                n0 = n - int(has_bias)
                wb = torch.empty(S, m, n0, dtype=torch.float32, device=dev)
                bb = torch.empty(S, m, dtype=torch.float32, device=dev) if has_bias else None
                w_mean = self.model_state_of(layer, 'weight').view(m, n0)
                b_mean = self.model_state_of(layer, 'bias').view(m, 1) if has_bias else None
                for k in range(S):
                    Vs_t = V[:, k * m:(k + 1) * m].t()                            # (m, n) view of V_s^T: K-contiguous columns
                    stage_b.append(ops.Gemm(second, Vs_t[:, :n0], wb[k], epilogue=ops.EPI_ADD_E, E=w_mean, tri=ops.TRI_A_LOWER))
                    if has_bias:
                        stage_b.append(ops.Gemm(second, Vs_t[:, n0:], bb[k].view(m, 1), epilogue=ops.EPI_ADD_E, E=b_mean,
                                                tri=ops.TRI_A_LOWER))
                weights[layer], biases[layer] = wb, bb
                keep += [zt, V]
            stage_a.sort(key=lambda j: -(j.A.shape[0] * j.A.shape[1] * j.B.shape[1]))
            stage_b.sort(key=lambda j: -(j.A.shape[0] * j.A.shape[1] * j.B.shape[1]))
            plan = (flat, ops.GemmPlan(stage_a), ops.GemmPlan(stage_b), SampleBank(S, weights, biases), keep)
            cache[key] = plan
        if plan[0] is not None:
            self._randn(plan[0].numel(), device=plan[0].device, out=plan[0])
        plan[1].run()
        plan[2].run()
        return plan[3]


class _InProjection:
    """``in_proj_weight`` / ``in_proj_bias`` of an nn.MultiheadAttention presented as a layer with ``weight`` / ``bias``
    (a `SampleBank` key; one object per module, so the key is stable)."""

    def __init__(self, mha: Module):
        self.mha = mha

    @classmethod
    def of(cls, mha: Module) -> "_InProjection":
        if "_curv_in_projection" not in mha.__dict__:
            mha.__dict__["_curv_in_projection"] = cls(mha)
        return mha.__dict__["_curv_in_projection"]

    weight = property(lambda self: self.mha.in_proj_weight)
    bias = property(lambda self: self.mha.in_proj_bias)

    @property
    def _parameters(self):
        return {'weight': self.mha.in_proj_weight, 'bias': self.mha.in_proj_bias}


class SampleBank:
    """`count` sampled parameter sets of an estimator's own layers (`KFAC.sample_many`): ``weights[layer]`` is
    (count, out, in[*kh*kw]) and ``biases[layer]`` (count, out) or None, each entry ``mean + sample``.  The buffers belong
    to the estimator's launch plan: the next `sample_many` call of the same size overwrites them."""

    def __init__(self, count: int, weights: Dict[Module, Tensor], biases: Dict[Module, Optional[Tensor]]):
        self.count, self.weights, self.biases = count, weights, biases


def _arena(shapes: Sequence[Sequence[int]], device, zero: bool = False):
    """One flat fp32 buffer and one view per shape, laid out back to back: whole-model elementwise steps
    (noise scaling, scalar-hyper-parameter inverts) then take ONE launch over the flat buffer instead of one
    per layer."""
    sizes = []
    for shape in shapes:
        count = 1
        for d in shape:
            count *= int(d)
        sizes.append(count)
    flat = (torch.zeros if zero else torch.empty)(max(sum(sizes), 1), dtype=torch.float32, device=device)
    views, pos = [], 0
    for shape, count in zip(shapes, sizes):
        views.append(flat[pos:pos + count].view(*shape))
        pos += count
    return flat, views


def _is_arena(flat: Optional[Tensor], tensors: Sequence[Tensor]) -> bool:
    """True if `tensors` are exactly the consecutive contiguous views `_arena` handed out for `flat`."""
    if flat is None:
        return False
    pos = flat.data_ptr()
    for t in tensors:
        if not t.is_contiguous() or t.dtype != torch.float32 or t.data_ptr() != pos:
            return False
        pos += 4 * t.numel()
    return pos <= flat.data_ptr() + 4 * flat.numel()


class EFB(Curvature):
    """Eigenvalue-corrected Kronecker factorisation (curvatures.py:395-460).

    ``state[layer]`` = Lambda (m, n) accumulating (U_G^T grad U_A)**2, ``diags[layer]`` the diagonal Fisher
    grad**2 * batch_size, ``eigvecs[layer] = (U_A, U_G)``, ``inv_state[layer] = (s Lambda + n)^-1/2``.

    Keyword-only extensions of the reference constructor: `shard` (this rank decomposes, updates, inverts and
    samples only the layers it owns; with a layer-sharded KFAC `factors` already holds just those) and
    `eigvecs` (a precomputed ``{layer: (U_A, U_G)}``, e.g. another estimator's, instead of decomposing)."""

    _mha_as_projections = True

    def __init__(self, model: Union[Module, Sequential], factors: Dict[Module, Tensor],
                 layer_types: Union[List[str], str] = None, *, shard=None, eigvecs=None):
        super().__init__(model, layer_types, shard=shard)
        if eigvecs is None:
            from .utils import get_eigenvectors
            if shard is not None:
                mine = {l for _, l in self._owned()}
                factors = {l: f for l, f in factors.items() if l in mine}
            eigvecs = get_eigenvectors(factors)
        self.eigvecs = eigvecs
        self.diags = dict()

    def _mine(self) -> List[Module]:
        """Owned layers that have eigenvectors, in ``modules()`` order."""
        return [l for _, l in self._owned() if l in self.eigvecs]

    def update(self, batch_size: int):
        layers = self._mine()
        if not layers:
            return
        grads = []
        for layer in layers:
            gw = layer.weight.grad
            if gw is None:
                raise RuntimeError("EFB.update: a selected layer has no gradient (call backward() first)")
            grads.append((gw.contiguous(), layer.bias.grad if layer.bias is not None else None))
        dev = grads[0][0].device
        missing = [k for k, layer in enumerate(layers) if layer not in self.state]
        if missing:
            # Lambda of all (new) layers in one zeroed arena: every update is then the same accumulate launch pair
            self._state_flat, views = _arena([(grads[k][0].shape[0], grads[k][0].numel() // grads[k][0].shape[0] +
                                               int(grads[k][1] is not None)) for k in missing], dev, zero=True)
            for k, v in zip(missing, views):
                self.state[layers[k]] = v
        # The launch plan is keyed by what it writes and by the eigenvectors only.  The gradients are staged into an arena the
        # plan owns with one batched copy per call: after zero_grad(set_to_none=True) every backward pass allocates new
        # .grad tensors, so a plan keyed by their addresses would be rebuilt on every update() and would pin the old
        # gradients (a second copy of all of them) through its descriptors.
        key = tuple(self.state[l].data_ptr() for l in layers) + tuple(t.data_ptr() for l in layers for t in self.eigvecs[l])
        plan = getattr(self, "_update_plan", None)
        if plan is None or plan[0] != key:
            stage1, stage2, staged = [], [], []
            _, tmps = _arena([tuple(self.state[l].shape) for l in layers], dev)
            shapes = []
            for layer, (gw, gb) in zip(layers, grads):
                m = gw.shape[0]
                shapes.append((m, gw.numel() // m))
                if gb is not None:
                    shapes.append((m, 1))
            _, views = _arena(shapes, dev)
            vi = 0
            for layer, (gw, gb), tmp in zip(layers, grads, tmps):
                n0 = gw.numel() // gw.shape[0]
                U_A, U_G = self.eigvecs[layer]
                if gb is None and min(gw.shape[0], n0) >= 64:
                    # bias-free layer: both products in the "rows times rows" form the LDS-DMA GEMM kernel takes (every operand
                    # contiguous along the summation index) - the eigenvectors are constants, their transposes are kept:
                    #   T^T = U_A^T W.grad^T  (n0 x m),   Lambda += (U_G^T T)**2 = (U_G^T (T^T)^T)**2
                    U_At, U_Gt = self._eigvecs_t(layer)
                    tmp_t = tmp.view(-1)[:n0 * gw.shape[0]].view(n0, gw.shape[0])
                    stage1.append(ops.Gemm(U_At, views[vi].t(), tmp_t))
                    staged.append(views[vi])
                    vi += 1
                    stage2.append(ops.Gemm(U_Gt, tmp_t.t(), self.state[layer], beta=1.0, epilogue=ops.EPI_SQUARE))
                    continue
                stage1.append(ops.Gemm(U_G.t(), views[vi], tmp[:, :n0]))              # U_G^T [W.grad | b.grad]
                staged.append(views[vi])
                vi += 1
                if gb is not None:
                    stage1.append(ops.Gemm(U_G.t(), views[vi], tmp[:, n0:]))
                    staged.append(views[vi])
                    vi += 1
                stage2.append(ops.Gemm(tmp, U_A, self.state[layer], beta=1.0, epilogue=ops.EPI_SQUARE))   # Lambda += (. U_A)**2
            plan = (key, ops.GemmPlan(stage1), ops.GemmPlan(stage2), staged)
            self._update_plan = plan
        srcs = []
        for gw, gb in grads:
            srcs.append(gw.view(gw.shape[0], -1))
            if gb is not None:
                srcs.append(gb.contiguous().view(-1, 1))
        ops.CopyPlan(plan[3], srcs).run()
        plan[1].run()
        plan[2].run()
        done = ops.sq_accumulate_many([(gw, gb.contiguous() if gb is not None else None, self.diags.get(layer), None)
                                       for layer, (gw, gb) in zip(layers, grads)], batch_size)      # one launch
        for layer, st in zip(layers, done):
            self.diags[layer] = st

    def _eigvecs_t(self, layer):
        """(U_A^T, U_G^T) as contiguous tensors, formed once per layer (the eigenvectors are constants of the estimator)."""
        cache = self.__dict__.setdefault("_eigvecs_t_cache", {})
        U_A, U_G = self.eigvecs[layer]
        hit = cache.get(layer)
        if hit is None or hit[0] != (U_A.data_ptr(), U_G.data_ptr()):
            hit = ((U_A.data_ptr(), U_G.data_ptr()), U_A.t().contiguous(), U_G.t().contiguous())
            cache[layer] = hit
        return hit[1], hit[2]

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        gindex = self._global_index()
        layers = list(self.state.keys())
        values = [self.state[l] for l in layers]
        prev = [self.inv_state.get(l) for l in layers]
        if not (all(p is not None and p.shape == v.shape for p, v in zip(prev, values))
                and _is_arena(getattr(self, "_inv_flat", None), prev)):
            # inverse state of all layers in one arena, overwritten in place by later calls (stable addresses
            # keep the sampler's launch plan valid; the noise scaling is one launch over the flat buffer)
            self._inv_flat, views = _arena([tuple(v.shape) for v in values], values[0].device)
            for layer, v in zip(layers, views):
                self.inv_state[layer] = v
        state_flat = getattr(self, "_state_flat", None)
        if _is_scalar(add) and _is_scalar(multiply) and _is_arena(state_flat, values) and \
                sum(v.numel() for v in values) == self._inv_flat.numel() == state_flat.numel():
            # one pair of hyper-parameters for every layer and both dicts are whole arenas: one launch
            ops.rsqrt_affine(state_flat, float(add), float(multiply), out=self._inv_flat)
            return
        for position, (layer, value) in enumerate(zip(layers, values)):
            n, s = self._hyper(add, multiply, gindex.get(layer, position), max(len(gindex), len(layers)))
            ops.rsqrt_affine(value, n, s, out=self.inv_state[layer])

    def sample(self, layer: Module, z: Optional[Tensor] = None) -> Tensor:
        """(U_A (z * inv^T) U_G^T)^T = U_G (z^T * inv) U_A^T -> (m, n) (curvatures.py:453-460)."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        first, second = self.eigvecs[layer]
        lambdas = self.inv_state[layer]
        n, m = first.size(0), second.size(0)
        if z is None:
            z = self._randn(n, m, device=first.device)
        zt = ops.mul2d(z.t(), lambdas)                                    # (m, n)
        pt = torch.empty(n, m, dtype=torch.float32, device=first.device)
        out = torch.empty(m, n, dtype=torch.float32, device=first.device)
        ops.gemm_batched([ops.Gemm(first, zt.t(), pt)])                   # P^T = U_A zt^T (see sample_and_replace)
        ops.gemm_batched([ops.Gemm(second, pt.t(), out)])                 # U_G P
        return out

    def sample_and_replace(self, noise: Optional[Dict[Module, Tensor]] = None):
        """Fused form of the base-class loop (same result as curvatures.py:117-129 with EFB.sample): the
        scaled noise, then two batched GEMM launches for the whole model, the second writing
        ``mean + sample`` straight into the parameters.  `noise[layer]` (n, m) may be supplied.  The launches
        are described once and replayed while the tensors involved stay where they are."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        layers = self._mine()
        self._reload_mean(skip=[p for l in layers for p in (l.weight, l.bias) if p is not None])
        if not layers:
            self._allgather_sampled()
            return
        key = (noise is None, tuple(t.data_ptr() for l in layers for t in (*self.eigvecs[l], self.inv_state[l])),
               tuple(p.data_ptr() for l in layers for p in (l.weight, l.bias) if p is not None),
               tuple(self.model_state_of(l, nm).data_ptr() for l in layers for nm in ('weight', 'bias')
                     if getattr(l, nm) is not None))
        plan = self._sample_plans().get(key)
        if plan is None:
            dev = self.inv_state[layers[0]].device
            shapes = [tuple(self.inv_state[l].shape) for l in layers]               # (m, n)
            zflat, zts = _arena(shapes, dev)
            _, tmps = _arena(shapes, dev)
            stage1, stage2 = [], []
            for layer, zt, tmp in zip(layers, zts, tmps):
                first, second = self.eigvecs[layer]
                n, m = first.size(0), second.size(0)
                # U_G zt U_A^T as  P^T = U_A zt^T (n x m),  out = U_G P: both products "rows times rows" (every operand
                # contiguous along the summation index - the form the LDS-DMA GEMM kernel takes), no transposed copies
                pt = tmp.view(-1).view(n, m)
                stage1.append(ops.Gemm(first, zt.t(), pt))
                n0 = n - int(layer.bias is not None)
                w = layer.weight.data.view(m, n0)
                w_mean = self.model_state_of(layer, 'weight').view(m, n0)
                p = pt.t()                                                          # (m, n) view of P
                stage2.append(ops.Gemm(second, p[:, :n0], w, epilogue=ops.EPI_ADD_E, E=w_mean))
                if layer.bias is not None:
                    b = layer.bias.data.view(m, 1)
                    b_mean = self.model_state_of(layer, 'bias').view(m, 1)
                    stage2.append(ops.Gemm(second, p[:, n0:], b, epilogue=ops.EPI_ADD_E, E=b_mean))
            stage1.sort(key=lambda j: -(j.A.shape[0] * j.A.shape[1] * j.B.shape[1]))
            stage2.sort(key=lambda j: -(j.A.shape[0] * j.A.shape[1] * j.B.shape[1]))
            inv_flat = getattr(self, "_inv_flat", None)
            whole = _is_arena(inv_flat, [self.inv_state[l] for l in layers]) and \
                sum(self.inv_state[l].numel() for l in layers) == zflat.numel()
            plan = (key, zflat, zts, ops.GemmPlan(stage1), ops.GemmPlan(stage2), inv_flat if whole else None)
            self._keep_plan(key, plan)
        _, zflat, zts, plan1, plan2, inv_flat = plan
        if noise is None:
            # z^T of the reference drawn directly in (m, n) layout: the transpose of iid noise is iid noise
            self._randn(zflat.numel(), device=zflat.device, out=zflat)
            if inv_flat is not None:
                ops.mul(zflat, inv_flat[:zflat.numel()], out=zflat)                 # one launch for the model
            else:
                for layer, zt in zip(layers, zts):
                    ops.mul(zt, self.inv_state[layer], out=zt)
        else:
            for layer, zt in zip(layers, zts):
                ops.mul2d(noise[layer].t(), self.inv_state[layer], out=zt)          # (m, n)
        plan1.run()
        plan2.run()
        self._allgather_sampled()


class INF(Curvature):
    """Sparse information form: low-rank eigen subset + diagonal correction (curvatures.py:463-672).

    ``state[layer] = (U_A[:, I], U_G[:, J], lambda[I x J], D)``; ``inv_state[layer] = (U_A_lr, U_G_lr, r, P_c)``.
    The (n m) x (a b) Kronecker matrix V_s of the reference's pre_sampler is never formed: V_s^T V_s is
    computed in closed form from Khatri-Rao squares, and the dense chain after it,
    L_c = (C^-1 + vtv)^-1 with C = A^-T (B - I) A^-1, A = chol(vtv), B = chol(vtv + I), is evaluated as the
    algebraically identical A^-T (I - B^-1) A^-1 in fp64 (no symmetry is assumed; P_c stays non-symmetric).

    Keyword-only extensions of the reference constructor: `shard` (only this rank's layers are decomposed,
    reduced, inverted and sampled; dicts coming from sharded estimators already hold just those) and `eigvecs`
    (reuse e.g. ``efb.eigvecs`` instead of decomposing the same factors again, curvatures.py:403 vs :473)."""

    _mha_as_projections = True

    def __init__(self, model: Union[Module, Sequential], diags: Dict[Module, Tensor],
                 factors: Dict[Module, Tensor], lambdas: Dict[Module, Tensor],
                 layer_types: Union[List[str], str] = None, *, shard=None, eigvecs=None):
        super().__init__(model, layer_types, shard=shard)
        assert diags.keys() == factors.keys() == lambdas.keys()
        if shard is not None:
            mine = {l for _, l in self._owned()}
            diags = {l: v for l, v in diags.items() if l in mine}
            factors = {l: v for l, v in factors.items() if l in mine}
            lambdas = {l: v for l, v in lambdas.items() if l in mine}
        if eigvecs is None:
            from .utils import get_eigenvectors
            eigvecs = get_eigenvectors(factors)
        self.eigvecs = eigvecs
        self.lambdas = lambdas
        self.diags = diags

    def update(self, rank: int = 100):
        layers = list(self.diags.keys())
        # lambda^T.flatten() (index i*m + j) of every layer, then the index sets of all layers in one launch and
        # one read-back (a launch + host sync per layer was a quarter of update() on ResNet-18)
        vecs = {layer: ops.gather2d(self.lambdas[layer].t()).view(-1) for layer in layers}
        need = [layer for layer in layers if rank < vecs[layer].shape[0]]
        picked = dict(zip(need, ops.inf_select_many(
            [vecs[layer] for layer in need],
            [(self.eigvecs[layer][0].shape[0], self.eigvecs[layer][1].shape[0]) for layer in need], rank)))
        stage1, stage2 = [], []
        # Lambda_lr and D of all layers live in one arena each: invert() with one pair of hyper-parameters then clamps,
        # scales and inverts them in three launches instead of three per layer
        dev = vecs[layers[0]].device if layers else None
        shapes = []
        for layer in layers:
            n, m = self.eigvecs[layer][0].shape[0], self.eigvecs[layer][1].shape[0]
            a, b = (picked[layer][0].numel(), picked[layer][1].numel()) if layer in picked else (n, m)
            shapes.append((n, m, a, b))
        self._corr_flat, corrs = _arena([(n, m) for n, m, _, _ in shapes], dev) if layers else (None, [])
        self._lam_flat, lams = _arena([(a * b,) for _, _, a, b in shapes], dev) if layers else (None, [])
        for layer, corr, lam in zip(layers, corrs, lams):
            xxt_eigvecs, ggt_eigvecs = self.eigvecs[layer]
            n, m = xxt_eigvecs.shape[0], ggt_eigvecs.shape[0]
            lambda_vec = vecs[layer]
            diag_vec = ops.gather2d(self.diags[layer].t())                          # (n, m): index i*m + j
            if layer not in picked:
                ua, ug = xxt_eigvecs, ggt_eigvecs
                lam.copy_(lambda_vec)
            else:
                I, J = picked[layer]
                ua = ops.gather2d(xxt_eigvecs, cols=I)
                ug = ops.gather2d(ggt_eigvecs, cols=J)
                ops.gather2d(lambda_vec.view(n, m), rows=I, cols=J, out=lam)
            a, b = ua.shape[1], ug.shape[1]
            # D = diag_vec - ((U_A**2) Lambda_lr (U_G**2)^T).flatten()
            ua2, ug2 = ops.mul(ua, ua), ops.mul(ug, ug)
            tmp = torch.empty(n, b, dtype=torch.float32, device=ua.device)
            stage1.append(ops.Gemm(ua2, lam.view(a, b), tmp))
            stage2.append(ops.Gemm(tmp, ug2.t(), corr, alpha=-1.0, epilogue=ops.EPI_ADD_E, E=diag_vec))
            self.state[layer] = (ua, ug, lam, corr.view(-1))
        ops.gemm_batched(stage1)
        ops.gemm_batched(stage2)
        self.__dict__.pop("_sample_plan_cache", None)

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        # All layers advance together through the stages of pre_sampler (:538-572): one batched launch per GEMM
        # stage, ONE batched fp64 factorisation sweep for the 2 x layers matrices vtv and vtv + I (and one
        # status read-back) instead of a sweep and a host synchronisation per layer.
        layers, regs = list(self.state.keys()), []
        gindex = self._global_index()
        # r of all layers in one arena, overwritten in place by later calls (stable addresses for the sampler's plan)
        rs = [self.inv_state[l][2] for l in layers] if len(self.inv_state) == len(layers) else []
        if not rs or not _is_arena(getattr(self, "_r_flat", None), rs) or \
                any(r.numel() != self.state[l][3].numel() for r, l in zip(rs, layers)):
            self._r_flat, rs = _arena([(self.state[l][3].numel(),) for l in layers], self.state[layers[0]][3].device)
        hypers = [self._hyper(add, multiply, gindex.get(layer, position), max(len(gindex), len(layers)))
                  for position, layer in enumerate(layers)]
        one_pair = len(set(hypers)) == 1 and \
            _is_arena(getattr(self, "_corr_flat", None), [self.state[l][3] for l in layers]) and \
            _is_arena(getattr(self, "_lam_flat", None), [self.state[l][2] for l in layers]) and \
            self._corr_flat.numel() == self._r_flat.numel() == sum(self.state[l][3].numel() for l in layers) and \
            self._lam_flat.numel() == sum(self.state[l][2].numel() for l in layers)
        if one_pair:
            # the three elementwise steps of :521-530 over the arenas of update(): three launches for the whole model
            n, s = hypers[0]
            ops.clamp_min0_(self._corr_flat)                             # in place on `state`, like :523
            reg_flat = ops.sqrt_scale(self._lam_flat, s)
            ops.rsqrt_affine(self._corr_flat, n, s, out=self._r_flat)
            pos = 0
            for layer, r in zip(layers, rs):
                lr_frst_eigvecs, lr_scnd_eigvecs, lr_lambda, _ = self.state[layer]
                regs.append((lr_frst_eigvecs, lr_scnd_eigvecs, reg_flat[pos:pos + lr_lambda.numel()], r))
                pos += lr_lambda.numel()
        else:
            for (n, s), layer, r in zip(hypers, layers, rs):
                lr_frst_eigvecs, lr_scnd_eigvecs, lr_lambda, correction = self.state[layer]
                ops.clamp_min0_(correction)                              # in place on `state`, like :523
                reg_lr_lambda = ops.sqrt_scale(lr_lambda, s)
                ops.rsqrt_affine(correction, n, s, out=r)
                regs.append((lr_frst_eigvecs, lr_scnd_eigvecs, reg_lr_lambda, r))
        self._r_version = getattr(self, "_r_version", 0) + 1            # (the sampler's cached r**2 is stale)
        prev = [self.inv_state[l][3] if l in self.inv_state else None for l in layers]
        pre_samples = self.pre_sampler_many(regs, outs=prev)
        for layer, (ua, ug, _, r), pre_sample in zip(layers, regs, pre_samples):
            self.inv_state[layer] = (ua, ug, r, pre_sample)

    RHS_SWEEP_ABOVE = 16        # layers per call from which T is built by the sweep's right-hand side mode (pre_sampler_many)

    @staticmethod
    def pre_sampler_many(regs, outs: Optional[Sequence[Optional[Tensor]]] = None) -> List[Tensor]:
        """`pre_sampler` for a list of (U_A_lr, U_G_lr, sigma, r) tuples, stage by stage.  `outs`: previous
        P_c tensors, overwritten in place where the shape still fits."""
        if not regs:
            return []
        # bound the float64 scratch of a batch (PA, M, V4, vtv, the two inverses, T, L_c per layer): layers are
        # processed in groups of at most ~24 GB, far below the 288 GB of the device
        def scratch(reg):
            (n, a), (m, b) = reg[0].shape, reg[1].shape
            return 8.0 * (n * a * a / 2 + m * b * b / 2 + n * m + a * a * m / 2 + 6.0 * (a * b) ** 2)
        if len(regs) > 1:
            groups, cur, size = [], [], 0.0
            for k, reg in enumerate(regs):
                if cur and size + scratch(reg) > 24e9:
                    groups.append(cur)
                    cur, size = [], 0.0
                cur.append(k)
                size += scratch(reg)
            groups.append(cur)
            if len(groups) > 1:
                res: List[Optional[Tensor]] = [None] * len(regs)
                for g in groups:
                    part = INF.pre_sampler_many([regs[k] for k in g], [outs[k] for k in g] if outs is not None else None)
                    for k, t in zip(g, part):
                        res[k] = t
                return res
        # V_s^T V_s in closed form, fp64 end to end: with strongly varying r (e.g. invert(1, 1000) on a ResNet) it is a
        # badly conditioned weighted Gram matrix, and an fp32 evaluation caps P_c - and the samples - at ~1e-3
        # (measured on ResNet-50's stem: 1.7e-3; with fp64: at the level of the fp32 inputs)
        first, parts = [], []
        # r**2 in fp64: one launch for the model when the r of its layers lie back to back (invert()'s arena)
        rs = [reg[3] for reg in regs]
        r2_flat = None
        if len(rs) > 1 and all(t.is_contiguous() and t.dtype == torch.float32 for t in rs) and \
                all(rs[k + 1].data_ptr() == rs[k].data_ptr() + 4 * rs[k].numel() for k in range(len(rs) - 1)):
            total = sum(t.numel() for t in rs)
            if rs[0].untyped_storage().nbytes() >= 4 * (rs[0].storage_offset() + total):      # ... inside ONE allocation
                r2_flat = ops.square_f64(torch.as_strided(rs[0].reshape(-1), (total,), (1,)))
        pos = 0
        for ua, ug, sigma, r in regs:
            (n, a), (m, b) = ua.shape, ug.shape
            # distinct column pairs only (i <= k): (n, a (a + 1) / 2), (m, b (b + 1) / 2) - half the flops of the first
            # product, a quarter of the second, the same values
            PA, PG = ops.colpairs_sym(ua), ops.colpairs_sym(ug)
            r2 = (r2_flat[pos:pos + n * m] if r2_flat is not None else ops.square_f64(r)).view(n, m)
            pos += n * m
            first.append(ops.Gemm64(PA.t(), r2))
            parts.append((PG, sigma, a, b))
        Ms = ops.gemm_f64_batched(first)
        V4s = ops.gemm_f64_batched([ops.Gemm64(M, PG) for M, (PG, _, _, _) in zip(Ms, parts)])
        vtvs = [ops.inf_vtv_assemble_sym(V4.contiguous(), sigma, a, b) for V4, (_, sigma, a, b) in zip(V4s, parts)]
        del first, Ms, V4s, parts
        # float64, lower triangular: A^-1 = chol(vtv)^-1, then T = (I - B^-1) A^-1 = A^-1 - chol(vtv + I)^-1 A^-1 by forward
        # substitution INSIDE the second sweep (`rhs`): no explicit B^-1, no product with it (a sixth of the call's flops).
        # The status words are read at the END of this function: a read-back here would leave the GPU idle while the
        # launches below are described and enqueued (14 of 115 ms on ResNet-50)
        if len(vtvs) > INF.RHS_SWEEP_ABOVE:
            invA = ops.chol_factor_inverse(vtvs, [0.0] * len(vtvs), check=False)
            info = ops.chol_factor_inverse.last_info
            Ts = ops.chol_factor_inverse(vtvs, [1.0] * len(vtvs), check=False, rhs=invA, rhs_minus=True)
            info = torch.cat([info, ops.chol_factor_inverse.last_info])
        else:
            # few layers (a small model, a layer-sharded rank): such a sweep is bound by its chain, and the chain-bound
            # kernels have no right-hand side form - both inverses explicitly in ONE sweep, then the product (its epilogue
            # reads A^-1 as the E operand)
            mats, adds = [], []
            for v in vtvs:
                mats += [v, v]
                adds += [0.0, 1.0]
            both = ops.chol_factor_inverse(mats, adds, check=False)
            info = ops.chol_factor_inverse.last_info
            invA = [both[2 * i] for i in range(len(vtvs))]
            Ts = [torch.empty_like(a) for a in invA]
            ops.gemm_f64_batched([ops.Gemm64(both[2 * i + 1], invA[i], T, alpha=-1.0, beta=1.0, E=invA[i],
                                             tri=ops.TRI64_A_LOWER | ops.TRI64_B_LOWER) for i, T in enumerate(Ts)])
        inv = [None] * (2 * len(regs))
        for i, a in enumerate(invA):
            inv[2 * i] = a
        out = []
        for i, (_, _, sigma, _) in enumerate(regs):
            prev = outs[i] if outs is not None else None
            shape = tuple(Ts[i].shape)
            if prev is None or tuple(prev.shape) != shape or prev.dtype != torch.float32 or prev.device != Ts[i].device \
                    or not prev.is_contiguous():
                prev = torch.empty(shape, dtype=torch.float32, device=Ts[i].device)
            out.append(prev)
        ops.gemm_f64_batched([ops.Gemm64(inv[2 * i].t(), T, tri=ops.TRI64_A_UPPER | ops.TRI64_B_LOWER, out32=out[i],
                                         row_scale=regs[i][2].contiguous(), col_scale=regs[i][2].contiguous())
                              for i, T in enumerate(Ts)])
        ops.check_factor_inverse_info(info)                              # RuntimeError where curvatures.py:566-567 raises: inside invert()
        return out

    @staticmethod
    def vtv(frst_eigvecs: Tensor, scnd_eigvecs: Tensor, reg_lambda: Tensor, reg_inv_correction: Tensor) -> Tensor:
        """V_s^T V_s, symmetrised, in closed form (no Kronecker matrix), evaluated in float64."""
        (n, a), (m, b) = frst_eigvecs.shape, scnd_eigvecs.shape
        PA, PG = ops.colpairs_sym(frst_eigvecs), ops.colpairs_sym(scnd_eigvecs)
        r2 = ops.square_f64(reg_inv_correction).view(n, m)
        M = ops.gemm_f64(PA.t(), r2)
        V4 = ops.gemm_f64(M, PG)
        return ops.inf_vtv_assemble_sym(V4.contiguous(), reg_lambda, a, b)

    @staticmethod
    def pre_sampler(frst_eigvecs: Tensor, scnd_eigvecs: Tensor, reg_lambda: Tensor,
                    reg_inv_correction: Tensor) -> Tensor:
        """P_c of one layer (curvatures.py:538-572)."""
        return INF.pre_sampler_many([(frst_eigvecs, scnd_eigvecs, reg_lambda, reg_inv_correction)])[0]

    def sample(self, layer: Module, X: Optional[Tensor] = None) -> Tensor:
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        a, b, c, d = self.inv_state[layer]
        return self.sampler(a, b, c, d, X=X, randn=self._randn).t()

    def sample_and_replace(self, noise: Optional[Dict[Module, Tensor]] = None):
        """The base-class loop (curvatures.py:117-129) with INF.sample, all layers advancing together through the
        five products of `sampler` (one batched launch each); the last one writes
        ``mean + (Y_l - Y_r)^T`` straight into the parameters through its output strides.  `noise[layer]`: the
        (n*m,) vector X of :578.  Launches are described once and replayed while the tensors stay in place."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        layers = [l for _, l in self._owned() if l in self.inv_state]
        self._reload_mean()                                   # the last product accumulates onto the mean
        if not layers:
            self._allgather_sampled()
            return
        key = (tuple(t.data_ptr() for l in layers for t in self.inv_state[l]),
               tuple(p.data_ptr() for l in layers for p in (l.weight, l.bias) if p is not None))
        plan = self._sample_plans().get(key)
        if plan is None:
            dev = self.inv_state[layers[0]][0].device
            dims = [(self.inv_state[l][0].shape, self.inv_state[l][1].shape) for l in layers]
            xflat, Xs = _arena([(n * m,) for (n, _), (m, _) in dims], dev)
            yflat, Ys = _arena([(n * m,) for (n, _), (m, _) in dims], dev)
            r2flat, r2s = _arena([(n, m) for (n, _), (m, _) in dims], dev)
            small = []
            for (n, a), (m, b) in dims:
                small += [(b, n), (a, b), (a * b, 1), (m, a)]
            _, sm = _arena(small, dev)
            stages = [[], [], [], [], []]
            for k, (layer, ((n, a), (m, b)), Y_l, r2) in enumerate(zip(layers, dims, Ys, r2s)):
                ua, ug, r, P = self.inv_state[layer]
                t1, xq_t, qx, t2 = sm[4 * k:4 * k + 4]
                stages[0].append(ops.Gemm(ug.t(), Y_l.view(m, n), t1))               # U_G^T unvec(Y_l): (b, n)
                stages[1].append(ops.Gemm(t1, ua, xq_t.t()))                         # flat order of Xq^T: k*b + l
                stages[2].append(ops.Gemm(P, xq_t.view(a * b, 1), qx))
                stages[3].append(ops.Gemm(ug, qx.view(b, a), t2))                    # U_G unvec(Qx): (m, a)
                # (Y_l - r^2 * (U_A t2^T)) as an (n, m) matrix is the transposed sample: rows < n0 go to the
                # weight seen through transposed strides, the last row to the bias (curvatures.py:67-82, 536)
                n0 = n - int(layer.bias is not None)
                Yv = Y_l.view(n, m)
                w_t = layer.weight.data.view(m, n0).t()
                stages[4].append(ops.Gemm(ua[:n0], t2.t(), w_t, alpha=-1.0, beta=1.0, epilogue=ops.EPI_MUL_E_ADD_F,
                                          E=r2[:n0], F=Yv[:n0]))
                if layer.bias is not None:
                    stages[4].append(ops.Gemm(ua[n0:], t2.t(), layer.bias.data.view(1, m), alpha=-1.0, beta=1.0,
                                              epilogue=ops.EPI_MUL_E_ADD_F, E=r2[n0:], F=Yv[n0:]))
            r_whole = _is_arena(getattr(self, "_r_flat", None), [self.inv_state[l][2] for l in layers]) and \
                sum(self.inv_state[l][2].numel() for l in layers) == xflat.numel()
            plan = (key, xflat, Xs, yflat, Ys, r2flat, r2s, [ops.GemmPlan(st) for st in stages],
                    self._r_flat if r_whole else None, [None])         # (last: the inversion whose r**2 `r2flat` holds)
            self._keep_plan(key, plan)
        _, xflat, Xs, yflat, Ys, r2flat, r2s, gemms, r_flat, r2_of = plan
        if noise is None:
            self._randn(xflat.numel(), device=xflat.device, out=xflat)
        else:
            ops.CopyPlan(Xs, [noise[l].reshape(-1).contiguous() for l in layers]).run()
        if r_flat is not None:                                  # Y_l = r * X and r^2 for the whole model
            rf = r_flat[:xflat.numel()]
            ops.mul(rf, xflat, out=yflat)
            if r2_of[0] != getattr(self, "_r_version", 0):       # r changes with invert() only: r^2 once per inversion
                ops.mul(rf, rf, out=r2flat)
                r2_of[0] = getattr(self, "_r_version", 0)
        else:
            for layer, X, Y_l, r2 in zip(layers, Xs, Ys, r2s):
                r = self.inv_state[layer][2]
                ops.mul(r, X, out=Y_l)
                ops.mul(r, r, out=r2.view(-1))
        for g in gemms:
            g.run()
        self._allgather_sampled()

    @staticmethod
    def sampler(frst_eigvecs: Tensor, scnd_eigvecs: Tensor, reg_inv_correction: Tensor, pre_sample: Tensor,
                X: Optional[Tensor] = None, randn=None) -> Tensor:
        """(Y_l - Y_r) as an (n, m) matrix (the reference returns it flat and reshapes in `sample`,
        curvatures.py:532-536, 574-600; reshape conventions reproduced literally)."""
        (n, a), (m, b) = frst_eigvecs.shape, scnd_eigvecs.shape
        dev = frst_eigvecs.device
        if X is None:
            X = randn(n * m, device=dev) if randn is not None else ops.randn((n * m,), dev, 0)
        Y_l = ops.mul(reg_inv_correction, X)                                   # (n*m,)
        t1 = torch.empty(b, n, dtype=torch.float32, device=dev)
        ops.gemm_batched([ops.Gemm(scnd_eigvecs.t(), Y_l.view(m, n), t1)])      # U_G^T unvec(Y_l): (b, n)
        xq_t = torch.empty(a, b, dtype=torch.float32, device=dev)               # Xq^T, so that its flat order is k*b + l
        ops.gemm_batched([ops.Gemm(t1, frst_eigvecs, xq_t.t())])
        qx = torch.empty(a * b, 1, dtype=torch.float32, device=dev)
        ops.gemm_batched([ops.Gemm(pre_sample, xq_t.view(a * b, 1), qx)])
        t2 = torch.empty(m, a, dtype=torch.float32, device=dev)
        ops.gemm_batched([ops.Gemm(scnd_eigvecs, qx.view(b, a), t2)])           # U_G unvec(Qx): (m, a)
        out = torch.empty(n, m, dtype=torch.float32, device=dev)
        r2 = ops.mul(reg_inv_correction, reg_inv_correction).view(n, m)
        # X_p_s = t2 U_A^T (m, n); Y_r[i*m + q] = r^2[i*m + q] X_p_s[q, i]: write X_p_s^T through the strides
        ops.gemm_batched([ops.Gemm(frst_eigvecs, t2.t(), out, alpha=-1.0, epilogue=ops.EPI_MUL_E_ADD_F, E=r2,
                                   F=Y_l.view(n, m))])
        return out
