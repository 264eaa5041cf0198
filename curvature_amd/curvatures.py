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

from . import ops

SUPPORTED_LAYERS = ['Linear', 'Conv2d', 'MultiheadAttention']


def _is_scalar(x) -> bool:
    """Python / numpy real scalars (a superset of what the reference accepts)."""
    return isinstance(x, numbers.Real) or (hasattr(x, "ndim") and getattr(x, "ndim") == 0)


class Curvature(ABC):
    """Base class: layer selection, mean weights, `_replace`, `sample_and_replace`.

    Mirrors curvature/curvatures.py:17-129.  Layers are selected by class NAME in ``model.modules()``
    order; that order is the layer index used by per-layer ``add`` / ``multiply`` lists."""

    def __init__(self, model: Union[Module, Sequential], layer_types: Union[List[str], str] = None):
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
        # optional layer sharding across ranks (curvature_amd.sharding.Shard); None = own every layer
        self.shard = None
        # device-side noise generator of the samplers (Philox): advance `noise_offset` per draw
        self.noise_seed = int(torch.initial_seed()) & (2 ** 63 - 1)
        self.noise_offset = 0

    # ------------------------------------------------------------------ helpers
    def _layers(self) -> List[Module]:
        """Selected Linear / Conv2d layers in ``model.modules()`` order (curvatures.py:120-122)."""
        out = []
        for layer in self.model.modules():
            name = layer.__class__.__name__
            if name in self.layer_types:
                if name in ('Linear', 'Conv2d'):
                    out.append(layer)
                elif name == 'MultiheadAttention':
                    raise NotImplementedError
        return out

    def _owned(self):
        """[(global layer index, layer)] of the layers this rank owns (all of them without a shard)."""
        layers = self._layers()
        if self.shard is None:
            return list(enumerate(layers))
        return [(i, l) for i, l in enumerate(layers) if self.shard.owns(i)]

    def _allgather_sampled(self):
        """Multi-GPU: the single collective of the path, reassembling every layer's sampled parameters."""
        if self.shard is not None and self.shard.world > 1:
            self.shard.allgather_params([[p.data for p in (l.weight, l.bias) if p is not None] for l in self._layers()])

    @staticmethod
    def _hyper(add, multiply, index: int, count: int):
        """(n, s) of layer `index`: lists only when BOTH are non-scalars (curvatures.py:361-365)."""
        if not _is_scalar(add) and not _is_scalar(multiply):
            assert len(add) == len(multiply) == count
            return float(add[index]), float(multiply[index])
        return float(add), float(multiply)

    def _randn(self, *shape, device, out: Optional[Tensor] = None) -> Tensor:
        numel = 1
        for s in shape:
            numel *= int(s)
        out = ops.randn(shape, device, self.noise_seed, self.noise_offset, out=out)
        self.noise_offset += (numel + 3) // 4
        return out

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
            self._reload_plans = {}
        if not live or not live[0][1].is_cuda:
            self.model.load_state_dict(self.model_state)     # CPU models: torch plumbing, nothing to batch
            return
        # parameters may have been re-homed (.to(), ...) and model_state may have been reassigned
        ptrs = tuple(v.data_ptr() for _, v in live)
        means = tuple(self.model_state[k].data_ptr() for k, _ in live)
        key = (ptrs, means, tuple(sorted(t.data_ptr() for t in skip)))
        plan = self._reload_plans.get(key)
        if plan is None:
            self._reload_plans.clear()
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

    def sample_and_replace(self):
        """Reset to the mean weights, then add one posterior sample per selected layer (curvatures.py:117-129)."""
        self._reload_mean()
        for _, layer in self._owned():
            _sample = self.sample(layer)
            self._replace(_sample, layer.weight, layer.bias)
        self._allgather_sampled()


class Diagonal(Curvature):
    """Diagonal Fisher: state += grad**2 * batch_size (curvatures.py:132-193)."""

    def update(self, batch_size: int):
        for layer in self._layers():
            bias_grad = layer.bias.grad if layer.bias is not None else None
            self.state[layer] = ops.sq_accumulate(layer.weight.grad.contiguous(), bias_grad, batch_size,
                                                  self.state.get(layer))

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        for index, (layer, value) in enumerate(self.state.items()):
            # Diagonal uses lists when both are list/tuple (curvatures.py:183); same outcome as _hyper
            n, s = self._hyper(add, multiply, index, len(self.state))
            self.inv_state[layer] = ops.rsqrt_affine(value, n, s)

    def sample(self, layer: Union[Module, str], z: Optional[Tensor] = None) -> Tensor:
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        inv = self.inv_state[layer]
        if z is None:
            z = self._randn(*inv.shape, device=inv.device)
        return ops.mul(z, inv)


class KFAC(Curvature):
    """Kronecker-factored Fisher (curvatures.py:264-392).

    ``state[layer] = [A, G]`` (fp32, exactly symmetric), ``inv_state[layer] = (L_A, L_G)`` with
    L = chol_lower((sqrt(s) F + sqrt(n) I)^-1).  ``record[layer] = [input, grad_output]``: unlike the
    reference the recorded grad_output is NOT pre-multiplied by the batch size (curvatures.py:310); the
    factor N is folded into the scale of the G-side SYRK, which saves one pass over every gradient."""

    def __init__(self, model: Union[Module, Sequential], layer_types: Union[List[str], str] = None):
        super().__init__(model, layer_types)
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
                    raise NotImplementedError

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
        ops.kfac_accumulate(jobs, events=getattr(self, "_timing_events", None))

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        factors, adds, muls = [], [], []
        all_layers = self._layers()
        gindex = {l: i for i, l in enumerate(all_layers)}
        for layer, value in self.state.items():
            # layer index = position among the selected layers in modules() order (== enumerate(state)
            # of the reference, curvatures.py:360, when every layer is owned)
            n, s = self._hyper(add, multiply, gindex.get(layer, 0), len(all_layers))
            for factor in value:
                factors.append(factor)
                adds.append(n)
                muls.append(s)
        # outputs of the previous call are overwritten in place (stable addresses keep the cached launch
        # plan of sample_and_replace valid); RuntimeError if a damped factor is not positive definite
        prev = [t for layer in self.state.keys() for t in self.inv_state.get(layer, (None, None))]
        chols = ops.chol_inv_lower(factors, adds, muls, outs=prev)
        for index, layer in enumerate(self.state.keys()):
            self.inv_state[layer] = (chols[2 * index], chols[2 * index + 1])

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
        # the second GEMM stage overwrites weight and bias of every owned layer with mean + sample
        self._reload_mean(skip=[p for _, l in owned for p in (l.weight, l.bias) if p is not None])
        # The two GEMM launches are described once and replayed while the tensors involved stay where they
        # are (invert() rewrites inv_state in place): per call only the noise is drawn.
        key = (noise is None, tuple(t.data_ptr() for _, l in owned for t in self.inv_state[l]),
               tuple(p.data_ptr() for _, l in owned for p in (l.weight, l.bias) if p is not None),
               tuple(self.model_state_of(l, nm).data_ptr() for _, l in owned for nm in ('weight', 'bias')
                     if getattr(l, nm) is not None),
               tuple(z.data_ptr() for z in noise.values()) if noise is not None else ())
        plan = getattr(self, "_sample_plan", None)
        if plan is None or plan[0] != key:
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
            self._sample_plan = plan
        if plan[1] is not None:
            self._randn(plan[1].numel(), device=plan[1].device, out=plan[1])
        plan[2].run()
        plan[3].run()
        self._allgather_sampled()

class EFB(Curvature):
    """Eigenvalue-corrected Kronecker factorisation (curvatures.py:395-460).

    ``state[layer]`` = Lambda (m, n) accumulating (U_G^T grad U_A)**2, ``diags[layer]`` the diagonal Fisher
    grad**2 * batch_size, ``eigvecs[layer] = (U_A, U_G)``, ``inv_state[layer] = (s Lambda + n)^-1/2``."""

    def __init__(self, model: Union[Module, Sequential], factors: Dict[Module, Tensor],
                 layer_types: Union[List[str], str] = None):
        super().__init__(model, layer_types)
        from .utils import get_eigenvectors
        self.eigvecs = get_eigenvectors(factors)
        self.diags = dict()

    def update(self, batch_size: int):
        stage1, stage2 = [], []
        for layer in self._layers():
            gw = layer.weight.grad.contiguous()
            m = gw.shape[0]
            gw2 = gw.view(m, -1)
            n0 = gw2.shape[1]
            gb = layer.bias.grad if layer.bias is not None else None
            n = n0 + int(gb is not None)
            U_A, U_G = self.eigvecs[layer]
            tmp = torch.empty(m, n, dtype=torch.float32, device=gw.device)
            stage1.append(ops.Gemm(U_G.t(), gw2, tmp[:, :n0]))                      # U_G^T [W.grad | b.grad]
            if gb is not None:
                stage1.append(ops.Gemm(U_G.t(), gb.view(m, 1), tmp[:, n0:]))
            first = layer not in self.state
            if first:
                self.state[layer] = torch.empty(m, n, dtype=torch.float32, device=gw.device)
            stage2.append(ops.Gemm(tmp, U_A, self.state[layer], beta=0.0 if first else 1.0,
                                   epilogue=ops.EPI_SQUARE))                        # Lambda (+)= (. U_A)**2
            self.diags[layer] = ops.sq_accumulate(gw, gb, batch_size, self.diags.get(layer))
        ops.gemm_batched(stage1)
        ops.gemm_batched(stage2)

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        for index, (layer, value) in enumerate(self.state.items()):
            n, s = self._hyper(add, multiply, index, len(self.state))
            self.inv_state[layer] = ops.rsqrt_affine(value, n, s)

    def sample(self, layer: Module, z: Optional[Tensor] = None) -> Tensor:
        """(U_A (z * inv^T) U_G^T)^T = U_G (z^T * inv) U_A^T -> (m, n) (curvatures.py:453-460)."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        first, second = self.eigvecs[layer]
        lambdas = self.inv_state[layer]
        n, m = first.size(0), second.size(0)
        if z is None:
            z = self._randn(n, m, device=first.device)
        zt = ops.mul2d(z.t(), lambdas)                                    # (m, n)
        tmp = torch.empty(m, n, dtype=torch.float32, device=first.device)
        out = torch.empty(m, n, dtype=torch.float32, device=first.device)
        ops.gemm_batched([ops.Gemm(second, zt, tmp)])
        ops.gemm_batched([ops.Gemm(tmp, first.t(), out)])
        return out


    def sample_and_replace(self, noise: Optional[Dict[Module, Tensor]] = None):
        """Fused form of the base-class loop (same result as curvatures.py:117-129 with EFB.sample): the
        scaled noise per layer, then two batched GEMM launches for the whole model, the second writing
        ``mean + sample`` straight into the parameters.  `noise[layer]` (n, m) may be supplied."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        owned = self._owned()
        self._reload_mean(skip=[p for _, l in owned for p in (l.weight, l.bias) if p is not None])
        stage1, stage2 = [], []
        for _, layer in owned:
            first, second = self.eigvecs[layer]
            lambdas = self.inv_state[layer]
            n, m = first.size(0), second.size(0)
            z = noise[layer] if noise is not None else self._randn(n, m, device=first.device)
            zt = ops.mul2d(z.t(), lambdas)                                # (m, n)
            tmp = torch.empty(m, n, dtype=torch.float32, device=first.device)
            stage1.append(ops.Gemm(second, zt, tmp))
            n0 = n - int(layer.bias is not None)
            w = layer.weight.data.view(m, n0)
            w_mean = self.model_state_of(layer, 'weight').view(m, n0)
            ua_t = first.t()
            stage2.append(ops.Gemm(tmp, ua_t[:, :n0], w, epilogue=ops.EPI_ADD_E, E=w_mean))
            if layer.bias is not None:
                b = layer.bias.data.view(m, 1)
                b_mean = self.model_state_of(layer, 'bias').view(m, 1)
                stage2.append(ops.Gemm(tmp, ua_t[:, n0:], b, epilogue=ops.EPI_ADD_E, E=b_mean))
        ops.gemm_batched(stage1)
        ops.gemm_batched(stage2)
        self._allgather_sampled()


class INF(Curvature):
    """Sparse information form: low-rank eigen subset + diagonal correction (curvatures.py:463-672).

    ``state[layer] = (U_A[:, I], U_G[:, J], lambda[I x J], D)``; ``inv_state[layer] = (U_A_lr, U_G_lr, r, P_c)``.
    The (n m) x (a b) Kronecker matrix V_s of the reference's pre_sampler is never formed: V_s^T V_s is
    computed in closed form from Khatri-Rao squares, and the dense chain after it,
    L_c = (C^-1 + vtv)^-1 with C = A^-T (B - I) A^-1, A = chol(vtv), B = chol(vtv + I), is evaluated as the
    algebraically identical A^-T (I - B^-1) A^-1 in fp64 (no symmetry is assumed; P_c stays non-symmetric)."""

    def __init__(self, model: Union[Module, Sequential], diags: Dict[Module, Tensor],
                 factors: Dict[Module, Tensor], lambdas: Dict[Module, Tensor],
                 layer_types: Union[List[str], str] = None):
        super().__init__(model, layer_types)
        assert diags.keys() == factors.keys() == lambdas.keys()
        from .utils import get_eigenvectors
        self.eigvecs = get_eigenvectors(factors)
        self.lambdas = lambdas
        self.diags = diags

    def update(self, rank: int = 100):
        layers = list(self.diags.keys())
        # the index sets of all layers in one launch and one read-back (a launch + host sync per layer was a
        # quarter of update() on ResNet-18)
        vecs = {layer: self.lambdas[layer].t().contiguous().view(-1) for layer in layers}   # index i*m + j
        need = [layer for layer in layers if rank < vecs[layer].shape[0]]
        picked = dict(zip(need, ops.inf_select_many(
            [vecs[layer] for layer in need],
            [(self.eigvecs[layer][0].shape[0], self.eigvecs[layer][1].shape[0]) for layer in need], rank)))
        for layer in layers:
            xxt_eigvecs, ggt_eigvecs = self.eigvecs[layer]
            diags = self.diags[layer]
            n, m = xxt_eigvecs.shape[0], ggt_eigvecs.shape[0]
            lambda_vec = vecs[layer]
            diag_vec = diags.t().contiguous().view(-1)
            if layer not in picked:
                ua, ug, lam = xxt_eigvecs, ggt_eigvecs, lambda_vec
            else:
                I, J = picked[layer]
                ua = xxt_eigvecs.index_select(1, I).contiguous()
                ug = ggt_eigvecs.index_select(1, J).contiguous()
                lam = lambda_vec.view(n, m).index_select(0, I).index_select(1, J).contiguous().view(-1)
            a, b = ua.shape[1], ug.shape[1]
            # D = diag_vec - ((U_A**2) Lambda_lr (U_G**2)^T).flatten()
            ua2, ug2 = ops.mul(ua, ua), ops.mul(ug, ug)
            tmp = torch.empty(n, b, dtype=torch.float32, device=ua.device)
            corr = torch.empty(n, m, dtype=torch.float32, device=ua.device)
            ops.gemm_batched([ops.Gemm(ua2, lam.view(a, b), tmp)])
            ops.gemm_batched([ops.Gemm(tmp, ug2.t(), corr, alpha=-1.0, epilogue=ops.EPI_ADD_E, E=diag_vec.view(n, m))])
            self.state[layer] = (ua, ug, lam, corr.view(-1))

    def invert(self, add: Union[float, list, tuple] = 0., multiply: Union[float, list, tuple] = 1.):
        assert self.state, "State dict is empty. Did you call 'update' prior to this?"
        # All layers advance together through the stages of pre_sampler (:538-572): one batched launch per GEMM
        # stage, ONE batched fp64 factorisation sweep for the 2 x layers matrices vtv and vtv + I (and one
        # status read-back) instead of a sweep and a host synchronisation per layer.
        layers, regs = list(self.state.keys()), []
        for index, layer in enumerate(layers):
            n, s = self._hyper(add, multiply, index, len(self.state))
            lr_frst_eigvecs, lr_scnd_eigvecs, lr_lambda, correction = self.state[layer]
            ops.clamp_min0_(correction)                                  # in place on `state`, like :523
            reg_lr_lambda = ops.sqrt_scale(lr_lambda, s)
            reg_inv_correction = ops.rsqrt_affine(correction, n, s)
            regs.append((lr_frst_eigvecs, lr_scnd_eigvecs, reg_lr_lambda, reg_inv_correction))
        pre_samples = self.pre_sampler_many(regs)
        for layer, (ua, ug, _, r), pre_sample in zip(layers, regs, pre_samples):
            self.inv_state[layer] = (ua, ug, r, pre_sample)

    @staticmethod
    def pre_sampler_many(regs) -> List[Tensor]:
        """`pre_sampler` for a list of (U_A_lr, U_G_lr, sigma, r) tuples, stage by stage."""
        if not regs:
            return []
        dev = regs[0][0].device
        stage1, stage2, parts = [], [], []
        for ua, ug, sigma, r in regs:
            (n, a), (m, b) = ua.shape, ug.shape
            PA, PG = ops.colpairs(ua), ops.colpairs(ug)                  # (n, a*a), (m, b*b)
            r2 = ops.mul(r, r).view(n, m)
            M = torch.empty(a * a, m, dtype=torch.float32, device=dev)
            V4 = torch.empty(a * a, b * b, dtype=torch.float32, device=dev)
            stage1.append(ops.Gemm(PA.t(), r2, M))
            stage2.append(ops.Gemm(M, PG, V4))
            parts.append((V4, sigma, a, b))
        ops.gemm_batched(stage1)
        ops.gemm_batched(stage2)
        vtvs = [ops.inf_vtv_assemble(V4, sigma, a, b) for V4, sigma, a, b in parts]
        mats, adds = [], []
        for v in vtvs:
            mats += [v, v]
            adds += [0.0, 1.0]
        inv = ops.chol_factor_inverse(mats, adds)                        # float64, lower triangular
        out = []
        for i, (_, _, sigma, _) in enumerate(regs):
            A_inv, B_inv = inv[2 * i], inv[2 * i + 1]
            T = ops.gemm_f64(B_inv, A_inv, alpha=-1.0, beta=1.0, C=A_inv.clone())      # (I - B^-1) A^-1
            L_c = ops.gemm_f64(A_inv.t(), T)
            out.append(ops.diag_scale(L_c, sigma, sigma))
        return out

    @staticmethod
    def vtv(frst_eigvecs: Tensor, scnd_eigvecs: Tensor, reg_lambda: Tensor, reg_inv_correction: Tensor) -> Tensor:
        """V_s^T V_s, symmetrised, in closed form (no Kronecker matrix)."""
        (n, a), (m, b) = frst_eigvecs.shape, scnd_eigvecs.shape
        PA, PG = ops.colpairs(frst_eigvecs), ops.colpairs(scnd_eigvecs)      # (n, a*a), (m, b*b)
        r2 = ops.mul(reg_inv_correction, reg_inv_correction).view(n, m)
        M = torch.empty(a * a, m, dtype=torch.float32, device=PA.device)
        V4 = torch.empty(a * a, b * b, dtype=torch.float32, device=PA.device)
        ops.gemm_batched([ops.Gemm(PA.t(), r2, M)])
        ops.gemm_batched([ops.Gemm(M, PG, V4)])
        return ops.inf_vtv_assemble(V4, reg_lambda, a, b)

    @staticmethod
    def pre_sampler(frst_eigvecs: Tensor, scnd_eigvecs: Tensor, reg_lambda: Tensor,
                    reg_inv_correction: Tensor) -> Tensor:
        vtv = INF.vtv(frst_eigvecs, scnd_eigvecs, reg_lambda, reg_inv_correction)
        A_inv, B_inv = ops.chol_factor_inverse([vtv, vtv], [0.0, 1.0])        # float64, lower triangular
        T = ops.gemm_f64(B_inv, A_inv, alpha=-1.0, beta=1.0, C=A_inv.clone())      # (I - B^-1) A^-1
        L_c = ops.gemm_f64(A_inv.t(), T)
        return ops.diag_scale(L_c, reg_lambda, reg_lambda)

    def sample(self, layer: Module, X: Optional[Tensor] = None) -> Tensor:
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        a, b, c, d = self.inv_state[layer]
        return self.sampler(a, b, c, d, X=X, randn=self._randn).t()

    def sample_and_replace(self, noise: Optional[Dict[Module, Tensor]] = None):
        """The base-class loop (curvatures.py:117-129) with INF.sample, all layers advancing together through the
        five products of `sampler` (one batched launch each).  `noise[layer]`: the (n*m,) vector X of :578."""
        assert self.inv_state, "Inverse state dict is empty. Did you call 'invert' prior to this?"
        owned = self._owned()
        self._reload_mean()
        st1, st2, st3, st4, st5, outs = [], [], [], [], [], []
        for _, layer in owned:
            ua, ug, r, P = self.inv_state[layer]
            (n, a), (m, b) = ua.shape, ug.shape
            dev = ua.device
            X = noise[layer].reshape(-1) if noise is not None else self._randn(n * m, device=dev)
            Y_l = ops.mul(r, X)                                                    # (n*m,)
            t1 = torch.empty(b, n, dtype=torch.float32, device=dev)
            xq_t = torch.empty(a, b, dtype=torch.float32, device=dev)
            qx = torch.empty(a * b, 1, dtype=torch.float32, device=dev)
            t2 = torch.empty(m, a, dtype=torch.float32, device=dev)
            out = Y_l.clone().view(n, m)                                           # becomes Y_l - Y_r
            r2 = ops.mul(r, r).view(n, m)
            st1.append(ops.Gemm(ug.t(), Y_l.view(m, n), t1))                       # U_G^T unvec(Y_l): (b, n)
            st2.append(ops.Gemm(t1, ua, xq_t.t()))                                 # flat order of Xq^T: k*b + l
            st3.append(ops.Gemm(P, xq_t.view(a * b, 1), qx))
            st4.append(ops.Gemm(ug, qx.view(b, a), t2))                            # U_G unvec(Qx): (m, a)
            st5.append(ops.Gemm(ua, t2.t(), out, alpha=-1.0, beta=1.0, epilogue=ops.EPI_MUL_E, E=r2))
            outs.append(out)
        for stage in (st1, st2, st3, st4, st5):
            ops.gemm_batched(stage)
        for (_, layer), out in zip(owned, outs):
            self._replace(out.t(), layer.weight, layer.bias)
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
        out = Y_l.clone().view(n, m)                                           # becomes Y_l - Y_r
        r2 = ops.mul(reg_inv_correction, reg_inv_correction).view(n, m)
        # X_p_s = t2 U_A^T (m, n); Y_r[i*m + q] = r^2[i*m + q] X_p_s[q, i]: write X_p_s^T through the strides
        ops.gemm_batched([ops.Gemm(frst_eigvecs, t2.t(), out, alpha=-1.0, beta=1.0, epilogue=ops.EPI_MUL_E, E=r2)])
        return out
