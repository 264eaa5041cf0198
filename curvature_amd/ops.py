"""Tensor-level wrappers over the C ABI: torch tensors in, ``data_ptr()``s and shapes out.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in ``libcurv_hip.so``.
Every wrapper requires CUDA(=HIP) fp32 contiguous tensors and raises ``RuntimeError`` otherwise:
there is no CPU fallback.
"""
import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import curv_factor_desc, curv_inv_desc

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


def kfac_accumulate(jobs: Sequence[FactorJob]) -> None:
    """Grouped factor build over any number of factors: one SYRK launch + one reduce launch."""
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
    _lib.check(L.curv_kfac_accumulate(_lib.stream_ptr(), arr, n, ws.data_ptr(), ws.numel()),
               "curv_kfac_accumulate")


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
