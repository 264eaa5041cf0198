"""Tensor-level wrappers over the C ABI: torch tensors in, ``data_ptr()``s and shapes out.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in ``libcurv_hip.so``.
Every wrapper requires CUDA(=HIP) fp32 contiguous tensors and raises ``RuntimeError`` otherwise:
there is no CPU fallback.
"""
import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import curv_factor_desc, curv_gemm_desc, curv_inv_desc

_workspaces = {}


def _require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("curvature_amd runs on MI355X only: got a CPU tensor (no CPU fallback)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"curvature_amd expects float32 tensors, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("curvature_amd expects contiguous tensors")


def workspace(nbytes: int, device: torch.device, tag: str = "default") -> torch.Tensor:
    """A cached scratch buffer owned by torch (the library never allocates device memory)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


class FactorJob:
    """One Kronecker-factor accumulation: dst (+)= scale * unfold(src) unfold(src)^T."""
    __slots__ = ("src", "dst", "kernel", "stride", "padding", "has_bias", "scale", "first")

    def __init__(self, src, dst, kernel=(1, 1), stride=(1, 1), padding=(0, 0), has_bias=False,
                 scale=1.0, first=False):
        self.src, self.dst = src, dst
        self.kernel, self.stride, self.padding = tuple(kernel), tuple(stride), tuple(padding)
        self.has_bias, self.scale, self.first = bool(has_bias), float(scale), bool(first)


def kfac_accumulate(jobs: Sequence[FactorJob], events=None) -> None:
    """Grouped factor build over any number of factors: one SYRK launch + one reduce launch.

    `events` = (start, stop) handles from ``_lib.lib().curv_event_create()`` are recorded around the SYRK
    kernel (bench.py's roofline measurement)."""
    if not jobs:
        return
    n = len(jobs)
    arr = (curv_factor_desc * n)()
    for d, j in zip(arr, jobs):
        _require_gpu(j.src, j.dst)
        if j.src.dim() == 4:
            N, C, H, W = j.src.shape
        elif j.src.dim() == 2:
            (N, C), H, W = j.src.shape, 1, 1
        else:
            raise RuntimeError("factor source must be (N,C,H,W) or (N,C)")
        dim = C * j.kernel[0] * j.kernel[1] + int(j.has_bias)
        if tuple(j.dst.shape) != (dim, dim):
            raise RuntimeError(f"factor destination must be ({dim},{dim}), got {tuple(j.dst.shape)}")
        d.src, d.dst = j.src.data_ptr(), j.dst.data_ptr()
        d.N, d.C, d.H, d.W = N, C, H, W
        d.kh, d.kw = j.kernel
        d.sh, d.sw = j.stride
        d.ph, d.pw = j.padding
        d.has_bias, d.first, d.scale = int(j.has_bias), int(j.first), j.scale
    L = _lib.lib()
    need = L.curv_kfac_workspace_bytes(arr, n)
    if need == 0:
        _lib.check(2, "curv_kfac_workspace_bytes")
    ws = workspace(need, jobs[0].src.device, "kfac")
    if events is None:
        rc = L.curv_kfac_accumulate(_lib.stream_ptr(), arr, n, ws.data_ptr(), ws.numel())
    else:
        rc = L.curv_kfac_accumulate_timed(_lib.stream_ptr(), arr, n, ws.data_ptr(), ws.numel(), events[0], events[1])
    _lib.check(rc, "curv_kfac_accumulate")


def rsqrt_affine(value: torch.Tensor, add: float, multiply: float, out: Optional[torch.Tensor] = None):
    _require_gpu(value, out)
    if out is None:
        out = torch.empty_like(value)
    _lib.check(_lib.lib().curv_rsqrt_affine(_lib.stream_ptr(), value.data_ptr(), float(multiply), float(add),
                                            out.data_ptr(), value.numel()), "curv_rsqrt_affine")
    return out


def sq_accumulate(grad_w: torch.Tensor, grad_b: Optional[torch.Tensor], batch_size: float,
                  state: Optional[torch.Tensor]) -> torch.Tensor:
    """state (+)= batch_size * [grad_w.view(m,-1) | grad_b]**2 ; allocates when state is None."""
    _require_gpu(grad_w, grad_b, state)
    rows = grad_w.shape[0]
    cols_w = grad_w.numel() // rows
    first = state is None
    if first:
        state = torch.empty(rows, cols_w + (grad_b is not None), dtype=torch.float32, device=grad_w.device)
    _lib.check(_lib.lib().curv_sq_accumulate(_lib.stream_ptr(), grad_w.data_ptr(),
                                             grad_b.data_ptr() if grad_b is not None else None,
                                             rows, cols_w, float(batch_size), state.data_ptr(), int(first)),
               "curv_sq_accumulate")
    return state


def chol_inv_lower(factors: Sequence[torch.Tensor], adds: Sequence[float], multiplies: Sequence[float],
                   check: bool = True) -> List[torch.Tensor]:
    """[chol_lower((sqrt(s_i) F_i + sqrt(n_i) I)^-1)] for all factors in one batched sweep.

    Raises ``RuntimeError`` (like torch's cholesky in the reference, curvatures.py:378-380) when a damped
    factor is not positive definite; set ``check=False`` to skip the host read-back of the status words."""
    n = len(factors)
    if n == 0:
        return []
    arr = (curv_inv_desc * n)()
    outs = []
    for d, F, a, s in zip(arr, factors, adds, multiplies):
        _require_gpu(F)
        if F.dim() != 2 or F.shape[0] != F.shape[1]:
            raise RuntimeError("factor must be a square matrix")
        out = torch.empty_like(F)
        outs.append(out)
        d.F, d.L, d.n, d.add, d.multiply = F.data_ptr(), out.data_ptr(), F.shape[0], float(a), float(s)
    dev = factors[0].device
    info = torch.empty(n, dtype=torch.int32, device=dev)
    L = _lib.lib()
    need = L.curv_chol_inv_workspace_bytes(arr, n)
    ws = workspace(need, dev, "invert")
    _lib.check(L.curv_chol_inv_lower(_lib.stream_ptr(), arr, n, info.data_ptr(), ws.data_ptr(), ws.numel()),
               "curv_chol_inv_lower")
    if check:
        bad = torch.nonzero(info).flatten().tolist()
        if bad:
            raise RuntimeError(f"cholesky: damped factor(s) {bad} are not positive-definite "
                               f"(first failing pivot {int(info[bad[0]]) - 1})")
    return outs


EPI_NONE, EPI_SQUARE, EPI_MUL_E, EPI_ADD_E = 0, 1, 2, 3


class Gemm:
    """C = epilogue(alpha * A @ B) [+ beta * C] on 2-D views (any strides: .t() and slices are free)."""
    __slots__ = ("A", "B", "C", "E", "alpha", "beta", "epilogue")

    def __init__(self, A, B, C, alpha=1.0, beta=0.0, epilogue=EPI_NONE, E=None):
        self.A, self.B, self.C, self.E = A, B, C, E
        self.alpha, self.beta, self.epilogue = float(alpha), float(beta), int(epilogue)


def _check_view(t: torch.Tensor):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
        raise RuntimeError("GEMM operands must be 2-D float32 GPU tensors (no CPU fallback)")


def gemm_batched(jobs: Sequence[Gemm]) -> None:
    """All products in one launch (one work item per 64x64 output tile)."""
    if not jobs:
        return
    n = len(jobs)
    arr = (curv_gemm_desc * n)()
    for d, j in zip(arr, jobs):
        for t in (j.A, j.B, j.C):
            _check_view(t)
        M, K = j.A.shape
        K2, N = j.B.shape
        if K != K2 or tuple(j.C.shape) != (M, N):
            raise RuntimeError(f"GEMM shape mismatch: {tuple(j.A.shape)} @ {tuple(j.B.shape)} -> {tuple(j.C.shape)}")
        d.A, d.B, d.C = j.A.data_ptr(), j.B.data_ptr(), j.C.data_ptr()
        d.a_rs, d.a_cs = j.A.stride()
        d.b_rs, d.b_cs = j.B.stride()
        d.c_rs, d.c_cs = j.C.stride()
        if j.E is not None:
            _check_view(j.E)
            if tuple(j.E.shape) != (M, N):
                raise RuntimeError("GEMM epilogue operand must match the output shape")
            d.E = j.E.data_ptr()
            d.e_rs, d.e_cs = j.E.stride()
        d.M, d.N, d.K = M, N, K
        d.alpha, d.beta, d.epilogue = j.alpha, j.beta, j.epilogue
    L = _lib.lib()
    ws = workspace(L.curv_gemm_workspace_bytes(n), jobs[0].C.device, "gemm")
    _lib.check(L.curv_gemm_batched(_lib.stream_ptr(), arr, n, ws.data_ptr(), ws.numel()), "curv_gemm_batched")


def randn(shape, device, seed: int, offset: int = 0) -> torch.Tensor:
    """Standard normal noise from the library's Philox generator (counter `offset` in units of 4 values)."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _lib.check(_lib.lib().curv_randn(_lib.stream_ptr(), out.data_ptr(), out.numel(), int(seed) & (2 ** 64 - 1),
                                     int(offset)), "curv_randn")
    return out


def mul(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _require_gpu(a, b, out)
    if a.shape != b.shape:
        raise RuntimeError("mul: shape mismatch")
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_lib.lib().curv_mul(_lib.stream_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel()),
               "curv_mul")
    return out
