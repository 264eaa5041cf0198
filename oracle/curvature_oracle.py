"""CPU oracle for the KFAC / EFB / INF hot path of DLR-RM/curvature.

TEST INFRASTRUCTURE ONLY.  This module restates, in plain torch-CPU tensor arithmetic, what the
reference computes on its PyTorch-CPU path (file:line citations are relative to the reference
checkout).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package ``curvature_amd`` never does and has no CPU fallback.

Pinning: the reference has no tests or golden vectors for this path except the ``kron`` doctest
(curvature/utils.py:301-309).  The oracle is therefore pinned against OUTPUTS OF THE REFERENCE ITSELF,
generated in the build container by ``tools/make_golden.py`` (which imports ``/root/reference``) and
committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below
against those vectors.

All functions are dtype-generic: pass float32 tensors for the reference's own precision, float64
tensors for the high-precision twin used where the reference's fp32 LAPACK noise exceeds the 1e-4
parity tolerance.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# KFAC.update                                                     curvature/curvatures.py:312-352
# --------------------------------------------------------------------------------------------
def unfold_input(x: Tensor, kernel_size, stride, padding, has_bias: bool) -> Tensor:
    """Rows (c, kh, kw) [+ ones row], columns (n, oh, ow): the matrix the reference feeds to mm.

    Conv2d: F.unfold -> permute(1, 0, 2) -> view(n, N*L)   (curvatures.py:329-330);
    Linear: x.t()                                            (curvatures.py:332);
    bias:   a row of ones is appended                        (curvatures.py:333-335).
    """
    if x.dim() == 4:
        cols = F.unfold(x, kernel_size, padding=padding, stride=stride)       # (N, n0, L)
        mat = cols.permute(1, 0, 2).contiguous().view(cols.shape[1], -1)       # (n0, N*L)
    else:
        mat = x.t()
    if has_bias:
        mat = torch.cat([mat, torch.ones_like(mat[:1])], dim=0)
    return mat


def kfac_factors(x: Tensor, grad_out: Tensor, kernel_size=None, stride=None, padding=None,
                 has_bias: bool = True) -> Tuple[Tensor, Tensor]:
    """One layer's (A, G) contribution of one batch.

    ``grad_out`` is the RAW grad_output of the layer; the reference's backward hook multiplies it by
    the batch size (curvatures.py:310) before the outer product, reproduced here.
    A = X X^T / (N*L) (curvatures.py:336); G = g g^T / (N*L) with g = N * grad_out (:340-343).
    """
    X = unfold_input(x, kernel_size, stride, padding, has_bias)
    A = X @ X.t() / float(X.shape[1])
    g = grad_out * grad_out.size(0)
    if g.dim() == 4:
        g = g.permute(1, 0, 2, 3).contiguous().view(g.shape[1], -1)
    else:
        g = g.t()
    G = g @ g.t() / float(g.shape[1])
    return A, G


# --------------------------------------------------------------------------------------------
# KFAC.invert                                                     curvature/curvatures.py:354-385
# --------------------------------------------------------------------------------------------
def layer_hyper(add, multiply, index: int, n_layers: int) -> Tuple[float, float]:
    """(n, s) of layer `index`: lists are used only when BOTH are non-scalars (curvatures.py:361-365)."""
    if not isinstance(add, (float, int)) and not isinstance(multiply, (float, int)):
        assert len(add) == len(multiply) == n_layers
        return add[index], multiply[index]
    return float(add), float(multiply)


def damp(factor: Tensor, add: float, multiply: float) -> Tensor:
    """sqrt(s) F + sqrt(n) I, symmetrised (curvatures.py:368-375)."""
    reg = multiply ** 0.5 * factor + torch.diag(factor.new_full((factor.shape[0],), add ** 0.5))
    return (reg + reg.t()) / 2.0


def chol_of_inverse(reg: Tensor) -> Tensor:
    """Lower Cholesky factor of reg^-1 (the reference's ``reg.inverse().cholesky()``, :378-379)."""
    return torch.linalg.cholesky(torch.linalg.inv(reg))


def kfac_invert(A: Tensor, G: Tensor, add: float, multiply: float) -> Tuple[Tensor, Tensor]:
    return chol_of_inverse(damp(A, add, multiply)), chol_of_inverse(damp(G, add, multiply))


# --------------------------------------------------------------------------------------------
# KFAC.sample / _replace                               curvature/curvatures.py:387-392, 67-82
# --------------------------------------------------------------------------------------------
def kfac_sample(L_A: Tensor, L_G: Tensor, z: Tensor) -> Tensor:
    """(L_A z L_G^T)^T -> (m, n) with z of shape (n, m) (curvatures.py:391-392)."""
    return (L_A @ z @ L_G.t()).t()


def replace(sample: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tuple[Tensor, Optional[Tensor]]:
    """New (weight, bias) = mean + sample; the last column of `sample` is the bias (curvatures.py:77-82)."""
    new_bias = None
    if bias is not None:
        new_bias = bias + sample[:, -1].contiguous().view(*bias.shape)
        sample = sample[:, :-1]
    return weight + sample.contiguous().view(*weight.shape), new_bias


# --------------------------------------------------------------------------------------------
# get_eigenvectors / kron                                 curvature/utils.py:45-60, 288-310
# --------------------------------------------------------------------------------------------
def eigenvectors(factor: Tensor) -> Tensor:
    """Eigenvectors (columns, eigenvalues ascending) of F + F^T, upper triangle (utils.py:55-58)."""
    return torch.linalg.eigh(factor + factor.t(), UPLO="U")[1]


def kron(a: Tensor, b: Tensor) -> Tensor:
    """Standard Kronecker product (utils.py:310)."""
    return torch.einsum("ab,cd->acbd", a, b).contiguous().view(a.size(0) * b.size(0), a.size(1) * b.size(1))


# --------------------------------------------------------------------------------------------
# Diagonal / EFB                                   curvature/curvatures.py:141-193, 414-460
# --------------------------------------------------------------------------------------------
def grad_matrix(grad_w: Tensor, grad_b: Optional[Tensor]) -> Tensor:
    """[W.grad.view(m, -1) | b.grad] (curvatures.py:424-426)."""
    g = grad_w.contiguous().view(grad_w.shape[0], -1)
    if grad_b is not None:
        g = torch.cat([g, grad_b.unsqueeze(dim=1)], dim=1)
    return g


def diag_update(grad_w: Tensor, grad_b: Optional[Tensor], batch_size) -> Tensor:
    """grads**2 * batch_size (curvatures.py:152 / :431-434)."""
    return grad_matrix(grad_w, grad_b) ** 2 * batch_size


def efb_update(U_A: Tensor, U_G: Tensor, grad_w: Tensor, grad_b: Optional[Tensor]) -> Tensor:
    """(U_G^T grad U_A)**2 (curvatures.py:427)."""
    return (U_G.t() @ grad_matrix(grad_w, grad_b) @ U_A) ** 2


def rsqrt_affine(value: Tensor, add, multiply) -> Tensor:
    """reciprocal(s*v + n).sqrt() (curvatures.py:188, :449, :526)."""
    return torch.reciprocal(multiply * value + add).sqrt()


def diag_sample(inv_state: Tensor, z: Tensor) -> Tensor:
    """z * inv_state (curvatures.py:193)."""
    return z * inv_state


def efb_sample(U_A: Tensor, U_G: Tensor, inv_lambda: Tensor, z: Tensor) -> Tensor:
    """(U_A (z * inv_lambda^T) U_G^T)^T with z of shape (n, m) (curvatures.py:457-460)."""
    return (U_A @ (z * inv_lambda.t()) @ U_G.t()).t()


# --------------------------------------------------------------------------------------------
# BlockDiagonal                                            curvature/curvatures.py:196-261
# --------------------------------------------------------------------------------------------
def block_grad_vector(grad_w: Tensor, grad_b: Optional[Tensor]) -> Tensor:
    """[W.grad.view(-1) ; b.grad]: all weights first, the biases at the END (curvatures.py:215-217) - not the
    [W | b] matrix layout of the other estimators."""
    g = grad_w.contiguous().view(-1)
    if grad_b is not None:
        g = torch.cat([g, grad_b])
    return g


def block_update(grad_w: Tensor, grad_b: Optional[Tensor], batch_size) -> Tensor:
    """ger(g, g) * batch_size (curvatures.py:218)."""
    g = block_grad_vector(grad_w, grad_b)
    return torch.ger(g, g) * batch_size


def block_invert(value: Tensor, add, multiply) -> Tensor:
    """(s * F + diag(n)).inverse().cholesky()  - lower factor (curvatures.py:252-253)."""
    reg = torch.diag(value.new(value.shape[0]).fill_(add))
    return torch.linalg.cholesky((multiply * value + reg).inverse())


def block_sample(inv_state: Tensor, z: Tensor, weight_shape, has_bias: bool = True) -> Tensor:
    """x = z @ L, weights reshaped, bias as the last column (curvatures.py:258-261).  The reference views the
    weight part with ``weight.shape`` and then concatenates along dim 1, which only works for Linear layers (a
    4-D Conv2d weight cannot be concatenated with the 2-D bias column: SURVEY.md section 2 row 6); the form here
    - weight part as (out, -1) - is the same thing for Linear and is what `_replace` needs for Conv2d."""
    x = z @ inv_state
    n_w = int(np.prod(weight_shape))
    w = x[:n_w].contiguous().view(weight_shape[0], -1)
    if not has_bias:
        return w
    return torch.cat([w, torch.unsqueeze(x[n_w:], dim=1)], dim=1)


# --------------------------------------------------------------------------------------------
# INF                                                      curvature/curvatures.py:487-672
# --------------------------------------------------------------------------------------------
def inf_select(lambda_vec: Tensor, m: int, rank: int) -> Tuple[np.ndarray, np.ndarray]:
    """Row/column index sets of the `rank` largest |lambda| (curvatures.py:617-634), 0-based, ascending.

    lambda_vec is indexed i*m + j (i: A side, j: G side).  The reference's 1-based float arithmetic
    ``int((idx - 1.) / m + 1.)`` equals integer floor division for n*m < 2**24.
    Ties among equal |lambda| are broken by the (unstable) sort in the reference; fixtures avoid them.
    """
    order = torch.argsort(-torch.abs(lambda_vec))[:rank].cpu().numpy().astype(np.int64)
    return np.unique(order // m), np.unique(order % m)


def inf_dim_reduction(U_A: Tensor, U_G: Tensor, lambda_vec: Tensor, rank: int):
    """Low-rank eigenvector subsets and the I x J block of lambda (curvatures.py:602-647)."""
    if rank >= lambda_vec.shape[0]:
        return U_A, U_G, lambda_vec, None, None
    m = U_G.shape[1]
    I, J = inf_select(lambda_vec, m, rank)
    cross = (I[:, None] * m + J[None, :]).reshape(-1)
    return U_A[:, I], U_G[:, J], lambda_vec[cross], I, J


def inf_diag(U_A_lr: Tensor, U_G_lr: Tensor, lambda_lr: Tensor) -> Tensor:
    """diag of (U_A (x) U_G) diag(lambda) (U_A (x) U_G)^T, index i*m + q (curvatures.py:649-672).

    Closed form of the reference's per-row Kronecker slabs: ((U_A**2) Lambda (U_G**2)^T).flatten().
    """
    a, b = U_A_lr.shape[1], U_G_lr.shape[1]
    return ((U_A_lr ** 2) @ lambda_lr.view(a, b) @ (U_G_lr ** 2).t()).reshape(-1)


def inf_update(U_A: Tensor, U_G: Tensor, lambdas: Tensor, diags: Tensor, rank: int):
    """state[layer] of INF.update (curvatures.py:498-507): (U_A_lr, U_G_lr, lambda_lr, D)."""
    lambda_vec = lambdas.t().contiguous().view(-1)
    diag_vec = diags.t().contiguous().view(-1)
    ua, ug, lam, I, J = inf_dim_reduction(U_A, U_G, lambda_vec, rank)
    return ua, ug, lam, diag_vec - inf_diag(ua, ug, lam), I, J


def inf_vtv(U_A_lr: Tensor, U_G_lr: Tensor, sigma: Tensor, r: Tensor) -> Tensor:
    """V_s^T V_s with V_s = (r[:, None] * kron(U_A, U_G)) @ diag(sigma), literally (curvatures.py:556-565)."""
    V_s = r.contiguous().view(-1, 1) * kron(U_A_lr, U_G_lr) @ torch.diag(sigma)
    vtv = V_s.t() @ V_s
    return (vtv + vtv.t()) / 2.0


def inf_pre_sampler_from_vtv(vtv: Tensor, sigma: Tensor) -> Tensor:
    """The dense chain after vtv (curvatures.py:566-570).  P_c is NOT symmetric; kept literal."""
    eye = torch.eye(sigma.shape[0], dtype=vtv.dtype)
    A_c_inv = torch.linalg.inv(torch.linalg.cholesky(vtv))
    B_c = torch.linalg.cholesky(vtv + eye)
    C = A_c_inv.t() @ (B_c - eye) @ A_c_inv
    L_c = torch.linalg.inv(torch.linalg.inv(C) + vtv)
    return torch.diag(sigma) @ L_c @ torch.diag(sigma)


def inf_invert(U_A_lr: Tensor, U_G_lr: Tensor, lambda_lr: Tensor, correction: Tensor, add, multiply):
    """INF.invert for one layer (curvatures.py:521-530) -> (clamped D, sigma, r, vtv, P_c)."""
    correction = correction.clone()
    correction[correction < 0] = 0
    sigma = (multiply * lambda_lr).sqrt()
    r = torch.reciprocal(multiply * correction + add).sqrt()
    vtv = inf_vtv(U_A_lr, U_G_lr, sigma, r)
    return correction, sigma, r, vtv, inf_pre_sampler_from_vtv(vtv, sigma)


def inf_sampler(U_A_lr: Tensor, U_G_lr: Tensor, r: Tensor, P_c: Tensor, X: Tensor) -> Tensor:
    """INF.sampler + the reshape in INF.sample, with the noise X (n*m) supplied (curvatures.py:532-600)."""
    n, m = U_A_lr.shape[0], U_G_lr.shape[0]
    a, b = U_A_lr.shape[1], U_G_lr.shape[1]
    Y_l = r * X
    unvec_Y_l = Y_l.reshape((m, n))
    Xq = U_G_lr.t() @ unvec_Y_l @ U_A_lr
    Qx = P_c @ Xq.t().contiguous().view(-1)
    unvec_Qx = Qx.reshape((b, a))
    X_p_s = U_G_lr @ unvec_Qx @ U_A_lr.t()
    Y_r = r ** 2 * X_p_s.t().contiguous().view(-1)
    return (Y_l - Y_r).reshape(n, m).t()


# --------------------------------------------------------------------------------------------
# Whole-model drivers (used by bench.py's cpu_baseline leg and by the model-level parity tests)
# --------------------------------------------------------------------------------------------
SUPPORTED = ("Linear", "Conv2d")


def selected_layers(model: torch.nn.Module, layer_types: Sequence[str] = SUPPORTED) -> List[torch.nn.Module]:
    """Layers in ``model.modules()`` order filtered by class NAME (curvatures.py:121, :298, :321-323)."""
    return [l for l in model.modules() if l.__class__.__name__ in layer_types]


def capture(model: torch.nn.Module, inputs: Tensor, labels: Optional[Tensor] = None, seed: int = 0):
    """One forward/backward; returns {layer: (input, raw grad_output)} and the logits.

    Labels default to a draw from Categorical(logits) (scripts/test.py:39-40)."""
    rec = {}
    handles = []
    for layer in selected_layers(model):
        def fwd(mod, inp, out):
            rec[mod] = [inp[0].detach(), None]
            out.register_hook(lambda g, mod=mod: rec[mod].__setitem__(1, g.detach()))
        handles.append(layer.register_forward_hook(fwd))
    logits = model(inputs)
    if labels is None:
        gen = torch.Generator(device="cpu").manual_seed(seed)
        probs = torch.softmax(logits.detach().float().cpu(), dim=1)
        labels = torch.multinomial(probs, 1, generator=gen).squeeze(1).to(logits.device)
    loss = F.cross_entropy(logits, labels)
    model.zero_grad()
    loss.backward()
    for h in handles:
        h.remove()
    return rec, logits.detach(), labels


def layer_geometry(layer: torch.nn.Module):
    if layer.__class__.__name__ == "Conv2d":
        return dict(kernel_size=layer.kernel_size, stride=layer.stride, padding=layer.padding)
    return dict(kernel_size=None, stride=None, padding=None)


def model_kfac_update(state: dict, model: torch.nn.Module, rec: dict) -> dict:
    """KFAC.update over a whole model (curvatures.py:321-350)."""
    for layer in selected_layers(model):
        x, g = rec[layer]
        A, G = kfac_factors(x, g, has_bias=layer.bias is not None, **layer_geometry(layer))
        if layer in state:
            state[layer][0] += A
            state[layer][1] += G
        else:
            state[layer] = [A, G]
    return state


def model_kfac_invert(state: dict, add, multiply) -> dict:
    inv = {}
    for index, (layer, (A, G)) in enumerate(state.items()):
        n, s = layer_hyper(add, multiply, index, len(state))
        inv[layer] = kfac_invert(A, G, n, s)
    return inv


def model_kfac_sample(inv_state: dict, model: torch.nn.Module, noise: Optional[dict] = None) -> dict:
    out = {}
    for layer in selected_layers(model):
        L_A, L_G = inv_state[layer]
        z = noise[layer] if noise is not None else torch.randn(L_A.size(0), L_G.size(0), dtype=L_A.dtype)
        out[layer] = kfac_sample(L_A, L_G, z)
    return out


def mc_fisher_kfac(model: torch.nn.Module, batches, samples: int, label_fn) -> dict:
    """The MC-Fisher outer loop of the reference's driver (scripts/factors.py:47-61) with KFAC: per batch one
    forward, then `samples` times {labels ~ Categorical(logits) (here: label_fn(logits, batch, sample)),
    backward(retain_graph=True), KFAC.update}.  Literal: the A side is rebuilt for every sample."""
    state = {}
    for b, images in enumerate(batches):
        rec, handles = {}, []
        for layer in selected_layers(model):
            def fwd(mod, inp, out):
                rec[mod] = [inp[0].detach(), None]
                out.register_hook(lambda g, mod=mod: rec[mod].__setitem__(1, g.detach()))
            handles.append(layer.register_forward_hook(fwd))
        logits = model(images)
        for h in handles:
            h.remove()
        for smp in range(samples):
            labels = label_fn(logits, b, smp)
            loss = F.cross_entropy(logits, labels)
            model.zero_grad()
            loss.backward(retain_graph=True)
            state = model_kfac_update(state, model, rec)
    return state
