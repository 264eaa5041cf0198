"""Tensor-level wrappers over the C ABI: torch tensors in, ``data_ptr()``s and shapes out.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in ``libcurv_hip.so``.
Every wrapper requires CUDA(=HIP) fp32 contiguous tensors and raises ``RuntimeError`` otherwise:
there is no CPU fallback.
"""
import ctypes
import os
import threading
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import curv_factor_desc, curv_gemm_desc, curv_inv_desc, curv_sq_desc

_workspaces = {}
_workspace_lock = threading.Lock()


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
    """A cached scratch buffer owned by torch (the library never allocates device memory).

    One buffer per (device, tag, current stream): launches on one stream are ordered, so they may share
    scratch; two estimators driven from different streams (or threads, each with its own current stream) get
    different buffers and cannot overwrite each other's slabs or descriptor tables mid-kernel.  The buffer is
    allocated on the stream it is keyed by, so the caching allocator's stream ownership matches its use, and
    a regrown buffer's predecessor is only recycled for later work of that same stream."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, tag, int(torch.cuda.current_stream(index).cuda_stream))
    with _workspace_lock:
        buf = _workspaces.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
            _workspaces[key] = buf
    if _POISON:
        buf.fill_(0xFF)
    return buf


# CURV_DEBUG_POISON=1 (diagnostics, tests/test_poisoned_workspace_gpu.py): every scratch buffer is filled with NaN bit patterns
# each time it is handed to the library, and no descriptor table is assumed to have survived in it - a kernel that reads
# scratch it has not written shows up as a non-finite result instead of depending on what the buffer held before
_POISON = os.environ.get("CURV_DEBUG_POISON", "0") not in ("", "0")
if _POISON:
    # ... and every GPU tensor that torch.empty / torch.empty_like hands out in this process starts as NaNs (integers: all
    # bits set): an output a kernel only partly writes cannot pass a test on what the allocator happened to return
    _torch_empty, _torch_empty_like = torch.empty, torch.empty_like

    def _poisoned(t):
        if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 0:
            if t.is_floating_point():
                t.fill_(float("nan"))
            elif t.dtype == torch.uint8:
                t.fill_(0xFF)
            elif t.dtype in (torch.int8, torch.int16, torch.int32, torch.int64):
                t.fill_(-1)
        return t

    torch.empty = lambda *a, **k: _poisoned(_torch_empty(*a, **k))
    torch.empty_like = lambda *a, **k: _poisoned(_torch_empty_like(*a, **k))


def release_workspaces() -> None:
    """Drop every cached scratch buffer (they are re-created on demand)."""
    with _workspace_lock:
        _workspaces.clear()
    _kfac_last.clear()


_kfac_last = {}                        # thread ident -> the "kfac" workspace its last curv_kfac_accumulate call used


class FactorJob:
    """One Kronecker-factor accumulation: dst (+)= scale * unfold(src) unfold(src)^T."""
    __slots__ = ("src", "dst", "kernel", "stride", "padding", "has_bias", "scale", "first", "path_hint")

    def __init__(self, src, dst, kernel=(1, 1), stride=(1, 1), padding=(0, 0), has_bias=False,
                 scale=1.0, first=False, path_hint=0):
        self.src, self.dst = src, dst
        self.kernel, self.stride, self.padding = tuple(kernel), tuple(stride), tuple(padding)
        self.has_bias, self.scale, self.first = bool(has_bias), float(scale), bool(first)
        self.path_hint = int(path_hint)            # _lib.PATH_*: which launch form the UNSHARDED model takes


def small_path_flop(dim: int, K: int) -> float:
    """Executed flops of one factor in the small launch form (32 x 32 blocks on and above the diagonal): the quantity
    CURV_SMALL_MAX_FLOP bounds (csrc/syrk_small.hip)."""
    nb = (int(dim) + 31) // 32
    return 2.0 * 1024.0 * (nb * (nb + 1) // 2) * float(K)


def _factor_descs(jobs: Sequence[FactorJob]):
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
        d.path_hint = getattr(j, "path_hint", 0)
    return arr


def kfac_path_for(geometries) -> int:
    """The launch form (``_lib.PATH_SMALL`` / ``_lib.PATH_GROUPED``) a factor build of exactly these factors takes on its
    own, decided by the library (curv_kfac_path_for: flops, factor count, slice length, workgroup count).
    `geometries`: one ``(N, C, H, W, kernel, stride, padding, has_bias)`` per factor of the UNSHARDED model; host only."""
    geometries = list(geometries)
    n = len(geometries)
    if n == 0:
        return _lib.PATH_GROUPED
    arr = (curv_factor_desc * n)()
    for d, (N, C, H, W, kernel, stride, padding, has_bias) in zip(arr, geometries):
        d.N, d.C, d.H, d.W = int(N), int(C), int(H), int(W)
        d.kh, d.kw = kernel
        d.sh, d.sw = stride
        d.ph, d.pw = padding
        d.has_bias, d.first, d.scale, d.path_hint = int(has_bias), 1, 1.0, 0
    return int(_lib.lib().curv_kfac_path_for(arr, n))


PLAN_INFO_FIELDS = 25                 # CURV_PLAN_INFO_FIELDS


def kfac_plan_flops(jobs: Sequence[FactorJob]) -> List[int]:
    """Multiply-add FLOPs the launch plan executes for each job (curv_kfac_plan_info, last field): dim (dim + 1) K for
    a symmetric product; the sum over its 29 shifted correlations for a 3x3 / stride 1 / pad 1 factor."""
    if not jobs:
        return []
    arr = _factor_descs(jobs)
    out = (ctypes.c_longlong * (PLAN_INFO_FIELDS * len(jobs)))()
    _lib.check(_lib.lib().curv_kfac_plan_info(arr, len(jobs), out), "curv_kfac_plan_info")
    return [int(out[PLAN_INFO_FIELDS * i + PLAN_INFO_FIELDS - 1]) for i in range(len(jobs))]


def kfac_accumulate(jobs: Sequence[FactorJob], events=None) -> None:
    """Grouped factor build over any number of factors: one SYRK launch + one reduce launch.

    `events` = (start, stop) handles from ``_lib.lib().curv_event_create()`` are recorded around the SYRK
    kernel (bench.py's roofline measurement)."""
    if not jobs:
        return
    n = len(jobs)
    arr = _factor_descs(jobs)
    L = _lib.lib()
    need = L.curv_kfac_workspace_bytes(arr, n)
    if need == 0:
        _lib.check(2, "curv_kfac_workspace_bytes")
    ws = workspace(need, jobs[0].src.device, "kfac")
    # the "kfac" workspaces are written by this function only: if this thread's previous call used this very buffer,
    # its head still holds the descriptor table of that call and unchanged argument blocks need no second upload
    # (a ResNet-50 update() is 26 of them)
    me = threading.get_ident()
    flags = _lib.KFAC_TABLE_RESIDENT if (_kfac_last.get(me) is ws and not _POISON) else 0
    _kfac_last[me] = ws
    ev0, ev1 = events if events is not None else (None, None)
    rc = L.curv_kfac_accumulate_ex(_lib.stream_ptr(), arr, n, ws.data_ptr(), ws.numel(), flags, ev0, ev1)
    if rc != 0:
        _kfac_last.pop(me, None)
    _lib.check(rc, "curv_kfac_accumulate")


def rsqrt_affine(value: torch.Tensor, add: float, multiply: float, out: Optional[torch.Tensor] = None):
    _require_gpu(value, out)
    if out is None:
        out = torch.empty_like(value)
    _lib.check(_lib.lib().curv_rsqrt_affine(_lib.stream_ptr(), value.data_ptr(), float(multiply), float(add),
                                            out.data_ptr(), value.numel()), "curv_rsqrt_affine")
    return out


def sq_accumulate(grad_w: torch.Tensor, grad_b: Optional[torch.Tensor], batch_size: float,
                  state: Optional[torch.Tensor], first: Optional[bool] = None) -> torch.Tensor:
    """state (+)= batch_size * [grad_w.view(m,-1) | grad_b]**2 ; allocates when state is None.  `first`: overwrite a
    given (preallocated, uninitialised) state instead of adding to it; default: only when it is allocated here."""
    _require_gpu(grad_w, grad_b, state)
    rows = grad_w.shape[0]
    cols_w = grad_w.numel() // rows
    if first is None:
        first = state is None
    if state is None:
        state = torch.empty(rows, cols_w + (grad_b is not None), dtype=torch.float32, device=grad_w.device)
    elif tuple(state.shape) != (rows, cols_w + (grad_b is not None)) or not state.is_contiguous():
        raise RuntimeError("sq_accumulate: state does not match the gradient's [W | b] shape")
    _lib.check(_lib.lib().curv_sq_accumulate(_lib.stream_ptr(), grad_w.data_ptr(),
                                             grad_b.data_ptr() if grad_b is not None else None,
                                             rows, cols_w, float(batch_size), state.data_ptr(), int(first)),
               "curv_sq_accumulate")
    return state


def sq_accumulate_many(items, batch_size: float) -> List[torch.Tensor]:
    """`sq_accumulate` for a list of (grad_w, grad_b or None, state or None, first or None) in ONE launch
    (curv_sq_accumulate_batched); returns the state tensors (allocated where None was given)."""
    items = list(items)
    if not items:
        return []
    arr = (curv_sq_desc * len(items))()
    states = []
    for d, (grad_w, grad_b, state, first) in zip(arr, items):
        _require_gpu(grad_w, grad_b, state)
        if not grad_w.is_contiguous() or (grad_b is not None and not grad_b.is_contiguous()):
            raise RuntimeError("sq_accumulate_many: gradients must be contiguous")
        rows = grad_w.shape[0]
        cols_w = grad_w.numel() // rows
        if first is None:
            first = state is None
        if state is None:
            state = torch.empty(rows, cols_w + (grad_b is not None), dtype=torch.float32, device=grad_w.device)
        elif tuple(state.shape) != (rows, cols_w + (grad_b is not None)) or not state.is_contiguous():
            raise RuntimeError("sq_accumulate_many: state does not match the gradient's [W | b] shape")
        states.append(state)
        d.grad_w, d.grad_b, d.state = grad_w.data_ptr(), (grad_b.data_ptr() if grad_b is not None else None), state.data_ptr()
        d.rows, d.cols_w, d.first = rows, cols_w, int(first)
    _lib.check(_lib.lib().curv_sq_accumulate_batched(_lib.stream_ptr(), arr, len(items), float(batch_size)),
               "curv_sq_accumulate_batched")
    return states


def chol_inv_lower(factors: Sequence[torch.Tensor], adds: Sequence[float], multiplies: Sequence[float],
                   check: bool = True, outs: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """[chol_lower((sqrt(s_i) F_i + sqrt(n_i) I)^-1)] for all factors in one batched sweep.

    Raises ``RuntimeError`` (like torch's cholesky in the reference, curvatures.py:378-380) when a damped
    factor is not positive definite; set ``check=False`` to skip the host read-back of the status words."""
    n = len(factors)
    if n == 0:
        return []
    arr = (curv_inv_desc * n)()
    given = list(outs) if outs is not None else [None] * n
    outs = []
    for d, F, a, s, out in zip(arr, factors, adds, multiplies, given):
        _require_gpu(F)
        if F.dim() != 2 or F.shape[0] != F.shape[1]:
            raise RuntimeError("factor must be a square matrix")
        if out is None or out.shape != F.shape or out.device != F.device or not out.is_contiguous() \
                or out.dtype != torch.float32:
            out = torch.empty_like(F)          # `outs` entries are reused when they fit (stable pointers)
        outs.append(out)
        d.F, d.L, d.n, d.add, d.multiply = F.data_ptr(), out.data_ptr(), F.shape[0], float(a), float(s)
    dev = factors[0].device
    info = torch.empty(n, dtype=torch.int32, device=dev)
    L = _lib.lib()
    need = L.curv_chol_inv_workspace_bytes(arr, n)
    ws = workspace(need, dev, "invert")
    early = check and not torch.cuda.is_current_stream_capturing() and os.environ.get("CURV_EARLY_STATUS", "1") != "0"
    if early:
        # the verdict travels to pinned host memory BEFORE the finalize passes (curv_chol_inv_lower_status): the host
        # waits for that copy only, and what it does next - raising, or preparing the sampler's launches - runs in the
        # shadow of the finalize passes instead of behind them (0.12 ms of idle GPU per ResNet-50 step otherwise)
        host, event = _status_box(n, dev)
        _lib.check(L.curv_chol_inv_lower_status(_lib.stream_ptr(), arr, n, info.data_ptr(), ws.data_ptr(), ws.numel(),
                                                host.data_ptr(), event), "curv_chol_inv_lower_status")
    else:
        _lib.check(L.curv_chol_inv_lower(_lib.stream_ptr(), arr, n, info.data_ptr(), ws.data_ptr(), ws.numel()),
                   "curv_chol_inv_lower")
    chol_inv_lower.last_info = info              # (kept for older callers; process-global: use the attribute below)
    outs = _ListWithInfo(outs)
    outs.info = info                             # check=False: the caller reads it later (check_chol_info)
    if early:
        _lib.check(L.curv_event_synchronize(event), "curv_event_synchronize")    # the one host wait of invert()
        check_chol_info(host)
    elif check:
        check_chol_info(info)
    return outs


_status_boxes = threading.local()


def _status_box(n: int, device):
    """Pinned host words and a HIP event for the early verdict of one `chol_inv_lower` call: one box per thread AND
    device (a HIP event belongs to the device that was current when it was created; recording it on a stream of another
    device is an invalid-handle error), grown on demand; a call waits for its own copy before it returns."""
    boxes = _status_boxes.__dict__.setdefault("boxes", {})
    key = torch.device(device).index
    if key is None:
        key = torch.cuda.current_device()
    box = boxes.get(key)
    if box is None or box[0].numel() < n:
        with torch.cuda.device(key):
            if box is not None:
                _lib.lib().curv_event_destroy(box[1])
            event = _lib.lib().curv_event_create()
        if not event:
            raise RuntimeError("curv_event_create failed")
        box = boxes[key] = (torch.empty(max(n, 256), dtype=torch.int32).pin_memory(), event)
    return box[0][:n], box[1]


class _ListWithInfo(list):
    """The list of inverse factors of one `chol_inv_lower` call, carrying that call's status words (`.info`)."""
    info = None


def check_chol_info(info: torch.Tensor) -> None:
    """Raise ``RuntimeError`` if a status word of a finished `chol_inv_lower` sweep reports a non-positive pivot."""
    host = info.cpu()
    bad = torch.nonzero(host).flatten().tolist()
    if bad:
        lost = [i for i in bad if int(host[i]) < 0]
        if lost:
            # status -1: a workgroup of chol_square_kernel gave up waiting for a tile from another workgroup of its
            # launch (bounded spin: all workgroups of a factor must be resident at once) - not a property of the matrix
            raise RuntimeError(f"cholesky: inter-workgroup hand-off timed out for factor(s) {lost} (the GPU could not keep "
                               "the sweep's workgroups resident: CU-masked / partitioned device or heavy contention); "
                               "set CURV_LATENCY_MAX=0 to use the per-step launches")
        raise RuntimeError(f"cholesky: damped factor(s) {bad} are not positive-definite "
                           f"(first failing pivot {int(host[bad[0]]) - 1})")


EPI_NONE, EPI_SQUARE, EPI_MUL_E, EPI_ADD_E, EPI_MUL_E_ADD_F = 0, 1, 2, 3, 4
TRI_NONE, TRI_A_LOWER, TRI_B_UPPER = 0, 1, 2


class Gemm:
    """C = epilogue(alpha * A @ B) [+ beta * C] on 2-D views (any strides: .t() and slices are free)."""
    __slots__ = ("A", "B", "C", "E", "F", "alpha", "beta", "epilogue", "tri")

    def __init__(self, A, B, C, alpha=1.0, beta=0.0, epilogue=EPI_NONE, E=None, tri=0, F=None):
        self.A, self.B, self.C, self.E, self.F = A, B, C, E, F
        self.alpha, self.beta, self.epilogue, self.tri = float(alpha), float(beta), int(epilogue), int(tri)


def _check_view(t: torch.Tensor):
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
        raise RuntimeError("GEMM operands must be 2-D float32 GPU tensors (no CPU fallback)")


def _gemm_descs(jobs: Sequence[Gemm]):
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
        if j.F is not None:
            _check_view(j.F)
            if tuple(j.F.shape) != (M, N):
                raise RuntimeError("GEMM epilogue operand must match the output shape")
            d.F = j.F.data_ptr()
            d.f_rs, d.f_cs = j.F.stride()
        d.M, d.N, d.K = M, N, K
        d.alpha, d.beta, d.epilogue, d.tri = j.alpha, j.beta, j.epilogue, j.tri
    return arr


def gemm_batched(jobs: Sequence[Gemm]) -> None:
    """All products in one launch (one work item per 64x64 output tile)."""
    if not jobs:
        return
    GemmPlan(jobs).run()


class GemmPlan:
    """A fixed list of products: descriptors are built once, `run()` only enqueues (one launch).  The
    operand tensors are kept alive by the plan; their contents may change between runs."""

    def __init__(self, jobs: Sequence[Gemm]):
        self.jobs = list(jobs)
        self.n = len(self.jobs)
        self.descs = _gemm_descs(self.jobs) if self.n else None

    def run(self) -> None:
        if not self.n:
            return
        L = _lib.lib()
        if getattr(self, "_ws", None) is None:
            # table + slabs of K-sliced products (underfilled launches), owned by the plan: nobody else writes there, so
            # the device copy of the descriptor table is uploaded by the first run only (a ResNet-50 sample was 6 upload
            # launches per call otherwise).  Slab space is shared scratch in effect - only this plan's launches use it.
            nbytes = int(L.curv_gemm_workspace_bytes_for(self.descs, self.n))
            self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.jobs[0].C.device)
            if _POISON:
                self._ws.fill_(0xFF)
            self._resident_on = None
        stream = _lib.stream_ptr()
        # (the table is written by launches on a stream: only replays on that same stream may rely on it)
        flags = _lib.GEMM_TABLE_RESIDENT if self._resident_on == stream else 0
        rc = L.curv_gemm_batched_ex(stream, self.descs, self.n, self._ws.data_ptr(), self._ws.numel(), flags)
        self._resident_on = stream if rc == 0 else None
        _lib.check(rc, "curv_gemm_batched")


def randn(shape, device, seed: int, offset: int = 0, out: Optional[torch.Tensor] = None,
          counter: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Standard normal noise from the library's Philox generator (counter `offset` in units of 4 values).
    `counter`: a one-element int64 GPU tensor that holds the stream position instead (read by the kernel, advanced by
    a second launch): the form a captured HIP graph needs, see `curvature_amd.graph`."""
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=device)
    else:
        _require_gpu(out)
    if counter is not None:
        if not counter.is_cuda or counter.dtype != torch.int64 or counter.numel() != 1:
            raise RuntimeError("randn: the device counter must be a one-element int64 GPU tensor")
        _lib.check(_lib.lib().curv_randn_counter(_lib.stream_ptr(), out.data_ptr(), out.numel(), int(seed) & (2 ** 64 - 1),
                                                 counter.data_ptr()), "curv_randn_counter")
        return out
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


class CopyPlan:
    """A fixed list of (dst, src) tensor pairs copied with one or a few launches (curv_copy_batched).

    The descriptor array is built once; `run()` only enqueues.  Every pair must be contiguous, of equal
    byte size and on the GPU; the tensors are kept alive by the plan."""

    def __init__(self, dsts: Sequence[torch.Tensor], srcs: Sequence[torch.Tensor]):
        if len(dsts) != len(srcs):
            raise RuntimeError("CopyPlan: list lengths differ")
        for t in (*dsts, *srcs):                 # any dtype: this is a byte copy
            if not t.is_cuda:
                raise RuntimeError("curvature_amd runs on MI355X only: got a CPU tensor (no CPU fallback)")
        self._keep = (list(dsts), list(srcs))
        self.n = len(dsts)
        self.descs = (_lib.curv_copy_desc * max(self.n, 1))()
        for i, (d, s) in enumerate(zip(dsts, srcs)):
            if not (d.is_contiguous() and s.is_contiguous()):
                raise RuntimeError("CopyPlan: tensors must be contiguous")
            nb = d.numel() * d.element_size()
            if nb != s.numel() * s.element_size() or d.dtype != s.dtype:
                raise RuntimeError("CopyPlan: size / dtype mismatch")
            self.descs[i].dst, self.descs[i].src, self.descs[i].bytes = d.data_ptr(), s.data_ptr(), nb

    def run(self) -> None:
        if self.n:
            _lib.check(_lib.lib().curv_copy_batched(_lib.stream_ptr(), self.descs, self.n), "curv_copy_batched")


# ---------------------------------------------------------------------------------------------- EFB / INF helpers
from ._lib import curv_cholinv_desc, curv_gemm64_desc, curv_select_desc  # noqa: E402


def mul2d(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Contiguous a * b for two strided 2-D float32 views of equal shape."""
    _check_view(a)
    _check_view(b)
    if a.shape != b.shape:
        raise RuntimeError("mul2d: shape mismatch")
    if out is None:
        out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    else:
        _require_gpu(out)
        if out.shape != a.shape:
            raise RuntimeError("mul2d: output shape mismatch")
    _lib.check(_lib.lib().curv_mul2d(_lib.stream_ptr(), a.data_ptr(), a.stride(0), a.stride(1), b.data_ptr(),
                                     b.stride(0), b.stride(1), out.data_ptr(), a.shape[0], a.shape[1]), "curv_mul2d")
    return out


def gather2d(src: torch.Tensor, rows: Optional[torch.Tensor] = None, cols: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Contiguous copy of the strided 2-D view `src`, optionally restricted to the int64 index lists `rows` /
    `cols` (``src.index_select(0, rows).index_select(1, cols).contiguous()`` in one launch; a transposed view
    as `src` gives ``src.t().contiguous()``)."""
    _check_view(src)
    for idx in (rows, cols):
        if idx is not None and (not idx.is_cuda or idx.dtype != torch.int64 or not idx.is_contiguous()):
            raise RuntimeError("gather2d: index lists must be contiguous int64 GPU tensors")
    R = src.shape[0] if rows is None else rows.numel()
    C = src.shape[1] if cols is None else cols.numel()
    if out is None:
        out = torch.empty(R, C, dtype=torch.float32, device=src.device)
    else:
        _require_gpu(out)
        if out.numel() != R * C:
            raise RuntimeError("gather2d: output size mismatch")
    _lib.check(_lib.lib().curv_gather2d(_lib.stream_ptr(), src.data_ptr(), src.stride(0), src.stride(1),
                                        rows.data_ptr() if rows is not None else None,
                                        cols.data_ptr() if cols is not None else None, out.data_ptr(), R, C),
               "curv_gather2d")
    return out


def clamp_min0_(v: torch.Tensor) -> torch.Tensor:
    _require_gpu(v)
    _lib.check(_lib.lib().curv_clamp_min0(_lib.stream_ptr(), v.data_ptr(), v.numel()), "curv_clamp_min0")
    return v


def sqrt_scale(v: torch.Tensor, s: float) -> torch.Tensor:
    _require_gpu(v)
    out = torch.empty_like(v)
    _lib.check(_lib.lib().curv_sqrt_scale(_lib.stream_ptr(), v.data_ptr(), float(s), out.data_ptr(), v.numel()),
               "curv_sqrt_scale")
    return out


def inf_select_many(lambda_vecs, dims, rank: int):
    """[(I, J), ...] int64 index tensors of INF._dim_reduction (exact integers, ascending) for several layers:
    one launch (a workgroup per layer) and ONE host read-back of all low-rank sizes."""
    if not lambda_vecs:
        return []
    dev = lambda_vecs[0].device
    k = len(lambda_vecs)
    counts = torch.zeros(k, 2, dtype=torch.int32, device=dev)
    d = (curv_select_desc * k)()
    Is, Js = [], []
    for i, (vec, (n, m)) in enumerate(zip(lambda_vecs, dims)):
        _require_gpu(vec)
        I = torch.empty(n, dtype=torch.int64, device=dev)
        J = torch.empty(m, dtype=torch.int64, device=dev)
        Is.append(I)
        Js.append(J)
        d[i].lambda_vec, d[i].I, d[i].J, d[i].counts = vec.data_ptr(), I.data_ptr(), J.data_ptr(), counts[i].data_ptr()
        d[i].n, d[i].m, d[i].rank = n, m, int(rank)
    _lib.check(_lib.lib().curv_inf_select(_lib.stream_ptr(), d, k), "curv_inf_select")
    sizes = counts.tolist()                    # host read-back: sizes of the low-rank factors
    return [(I[:a], J[:b]) for I, J, (a, b) in zip(Is, Js, sizes)]


def inf_select(lambda_vec: torch.Tensor, n: int, m: int, rank: int):
    """(I, J) of one layer (see inf_select_many)."""
    return inf_select_many([lambda_vec], [(n, m)], rank)[0]


def colpairs(U: torch.Tensor, f64: bool = False) -> torch.Tensor:
    """(n, a) -> (n, a*a) with out[p, i*a+k] = U[p,i] U[p,k]; `f64`: exact products in float64."""
    _require_gpu(U)
    n, a = U.shape
    out = torch.empty(n, a * a, dtype=torch.float64 if f64 else torch.float32, device=U.device)
    fn = _lib.lib().curv_colpairs_f64 if f64 else _lib.lib().curv_colpairs
    _lib.check(fn(_lib.stream_ptr(), U.data_ptr(), n, a, U.stride(0), out.data_ptr()), "curv_colpairs")
    return out


def colpairs_sym(U: torch.Tensor) -> torch.Tensor:
    """(n, a) -> (n, a (a + 1) / 2) float64 with out[p, t(i, k)] = U[p,i] U[p,k] for i <= k, t(i, k) = i a - i (i - 1) / 2 + k - i:
    the distinct columns of `colpairs`."""
    _require_gpu(U)
    n, a = U.shape
    out = torch.empty(n, a * (a + 1) // 2, dtype=torch.float64, device=U.device)
    _lib.check(_lib.lib().curv_colpairs_sym_f64(_lib.stream_ptr(), U.data_ptr(), n, a, U.stride(0), out.data_ptr()),
               "curv_colpairs_sym_f64")
    return out


def inf_vtv_assemble_sym(V4p: torch.Tensor, sigma: torch.Tensor, a: int, b: int) -> torch.Tensor:
    """vtv (ab x ab, float64) from the packed V4p (a (a + 1) / 2 x b (b + 1) / 2) of `colpairs_sym` operands."""
    _require_gpu(sigma)
    if not V4p.is_cuda or not V4p.is_contiguous() or V4p.dtype != torch.float64 \
            or V4p.shape != (a * (a + 1) // 2, b * (b + 1) // 2):
        raise RuntimeError("inf_vtv_assemble_sym: bad V4p")
    out = torch.empty(a * b, a * b, dtype=torch.float64, device=V4p.device)
    _lib.check(_lib.lib().curv_inf_vtv_assemble_sym_f64(_lib.stream_ptr(), V4p.data_ptr(), sigma.data_ptr(), a, b,
                                                        out.data_ptr()), "curv_inf_vtv_assemble_sym_f64")
    return out


def square_f64(v: torch.Tensor) -> torch.Tensor:
    """float64 v**2 of a float32 tensor (same shape)."""
    _require_gpu(v)
    out = torch.empty(v.shape, dtype=torch.float64, device=v.device)
    _lib.check(_lib.lib().curv_square_f64(_lib.stream_ptr(), v.data_ptr(), out.data_ptr(), v.numel()), "curv_square_f64")
    return out


def inf_vtv_assemble(V4: torch.Tensor, sigma: torch.Tensor, a: int, b: int) -> torch.Tensor:
    """vtv (ab x ab) from V4 (a*a x b*b), in V4's precision (float32 or float64)."""
    _require_gpu(sigma)
    if not V4.is_cuda or not V4.is_contiguous() or V4.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("inf_vtv_assemble: bad V4")
    out = torch.empty(a * b, a * b, dtype=V4.dtype, device=V4.device)
    fn = _lib.lib().curv_inf_vtv_assemble_f64 if V4.dtype == torch.float64 else _lib.lib().curv_inf_vtv_assemble
    _lib.check(fn(_lib.stream_ptr(), V4.data_ptr(), sigma.data_ptr(), a, b, out.data_ptr()), "curv_inf_vtv_assemble")
    return out


def diag_scale(src: torch.Tensor, dl: torch.Tensor, dr: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float32 out[i,j] = src[i,j] dl[i] dr[j]; src float32 or float64, contiguous.  `out` is reused when it is a
    contiguous float32 tensor of the right shape on the right device."""
    if not src.is_cuda or not src.is_contiguous() or src.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("diag_scale: bad source")
    _require_gpu(dl, dr)
    if out is None or out.shape != src.shape or out.dtype != torch.float32 or out.device != src.device \
            or not out.is_contiguous():
        out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().curv_diag_scale(_lib.stream_ptr(), src.data_ptr(), int(src.dtype == torch.float64),
                                          out.data_ptr(), dl.data_ptr(), dr.data_ptr(), src.shape[0], src.shape[1]),
               "curv_diag_scale")
    return out


def chol_factor_inverse(mats: Sequence[torch.Tensor], diag_adds: Sequence[float], check: bool = True,
                        rhs: Optional[Sequence[Optional[torch.Tensor]]] = None, rhs_minus: bool = False,
                        pivot_mins: Optional[Sequence[float]] = None) -> List[torch.Tensor]:
    """[chol_lower(M + d I)^-1] in float64 for symmetric float32 or float64 matrices (batched).  `check=False` leaves
    the status words on the device (``chol_factor_inverse.last_info``) for `check_factor_inverse_info`: a caller with
    more launches to enqueue does that first and synchronises once, at its end.  `rhs[i]` (lower triangular, float64):
    entry i of the result is chol_lower(M_i + d_i I)^-1 rhs[i] instead - the forward substitution runs inside the sweep, at
    the cost of the inverse it replaces (with `rhs_minus`: rhs[i] minus that); the right-hand side may be ANOTHER entry's
    result only across calls.  `pivot_mins[i]` > 0: a pivot at or below it counts as failed (the status word then holds the
    number of columns factorised before it, plus one: the numerical rank of a Gram matrix in general position)."""
    n = len(mats)
    arr = (curv_cholinv_desc * n)()
    outs = []
    if pivot_mins is not None:
        for d, pm in zip(arr, pivot_mins):
            d.pivot_min = float(pm)
    for d, M, da in zip(arr, mats, diag_adds):
        if not M.is_cuda or not M.is_contiguous() or M.dtype not in (torch.float32, torch.float64) or M.dim() != 2:
            raise RuntimeError("chol_factor_inverse: contiguous float32 / float64 GPU matrices expected")
        X = torch.empty(M.shape, dtype=torch.float64, device=M.device)
        d.M, d.X, d.n, d.diag_add = M.data_ptr(), X.data_ptr(), M.shape[0], float(da)
        d.m_is_f64 = int(M.dtype == torch.float64)
        R = rhs[len(outs)] if rhs is not None else None
        if R is not None:
            if not R.is_cuda or R.dtype != torch.float64 or R.shape != M.shape or not R.is_contiguous():
                raise RuntimeError("chol_factor_inverse: a right-hand side is a contiguous float64 GPU matrix of M's shape")
            d.R = R.data_ptr()
            d.r_minus = int(bool(rhs_minus))
        outs.append(X)
    dev = mats[0].device
    info = torch.empty(n, dtype=torch.int32, device=dev)
    L = _lib.lib()
    ws = workspace(L.curv_chol_factor_inverse_workspace_bytes(arr, n), dev, "invert")
    _lib.check(L.curv_chol_factor_inverse(_lib.stream_ptr(), arr, n, info.data_ptr(), ws.data_ptr(), ws.numel()),
               "curv_chol_factor_inverse")
    chol_factor_inverse.last_info = info
    if check:
        check_factor_inverse_info(info)
    return outs


def check_factor_inverse_info(info: torch.Tensor) -> None:
    """Raise like `chol_factor_inverse` does when a status word of the batch is non-zero (one host synchronisation)."""
    bad = torch.nonzero(info).flatten().tolist()
    if bad:
        raise RuntimeError(f"cholesky: matrix/matrices {bad} are not positive-definite")


def eigh(mats: Sequence[torch.Tensor], with_values: bool = False, max_sweeps: int = 0, tol: float = 0.0,
         allow_unconverged: bool = False, _project: bool = True):
    """Eigenvectors (columns, ascending eigenvalues) of symmetric float32 matrices, batched block-Jacobi.

    Raises ``RuntimeError`` when the iteration has not converged to `tol` within `max_sweeps` sweeps (default 60 sweeps;
    default tolerance off(A) <= 1e-8 ||A|| for matrices up to 1024 wide and 5e-6 ||A|| above - where the reference's
    fp32 LAPACK delivers 1e-5; an explicit `tol` applies to every matrix; the loop ends at convergence) unless
    `allow_unconverged` is set, in which case the last iterate is returned and ``eigh.converged`` is False."""
    from ._lib import curv_eigh_desc
    if len(mats) == 0:
        return []
    for F in mats:
        _require_gpu(F)
        if F.dim() != 2 or F.shape[0] != F.shape[1]:
            raise RuntimeError("eigh: square matrices expected")
    # Wide, numerically rank-deficient matrices (a KFAC factor with fewer samples than rows) are projected onto their range
    # inside the library (csrc/eigh_lowrank.hip: only the k x k projected problem is iterated on); the others - and every
    # matrix that turns out not to be rank-deficient - take the block-Jacobi iteration on the whole matrix.  `_project=False`
    # (or CURV_EIGH_LOWRANK=0 in the environment) asks for the plain iteration.
    n = len(mats)
    arr = (curv_eigh_desc * n)()
    vecs, vals = [], []
    mats = [F if (F.dtype == torch.float32 and F.is_contiguous()) else F.float().contiguous() for F in mats]
    for d, F in zip(arr, mats):
        U = torch.empty_like(F)
        w = torch.empty(F.shape[0], dtype=torch.float32, device=F.device)
        vecs.append(U)
        vals.append(w)
        d.F, d.U, d.w, d.n = F.data_ptr(), U.data_ptr(), w.data_ptr(), F.shape[0]
    L = _lib.lib()
    need = L.curv_syevd_workspace_bytes(arr, n)
    if need == 0:
        raise RuntimeError("eigh: matrix size out of range")
    ws = workspace(need, mats[0].device, "eigh")
    sweeps = ctypes.c_int(0)
    ranks = (ctypes.c_int * n)()
    # (an explicit sweep limit keeps the projection off: max_sweeps = 60 is the plain iteration's own default)
    plain = (not _project) and max_sweeps == 0 and tol <= 0.0
    rc = L.curv_syevd_ex(_lib.stream_ptr(), arr, n, ws.data_ptr(), ws.numel(), 60 if plain else int(max_sweeps), float(tol),
                         ctypes.byref(sweeps), ranks)
    eigh.last_ranks = {i: int(k) for i, k in enumerate(ranks) if k > 0}      # position in `mats` -> rank of the projected problem
    eigh.last_lowrank = len(eigh.last_ranks)
    eigh.last_sweeps = sweeps.value
    eigh.converged = rc == 0
    if not (rc == _lib.ERR_NOT_CONVERGED and allow_unconverged):
        _lib.check(rc, "curv_syevd")
    return (vecs, vals) if with_values else vecs



def kron(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Kronecker product, reference index convention (utils.py:288-310): (ar*br) x (ac*bc)."""
    a, b = a.contiguous(), b.contiguous()
    _require_gpu(a, b)
    if a.dim() != 2 or b.dim() != 2:
        raise RuntimeError("kron: 2-D operands expected")
    out = torch.empty(a.shape[0] * b.shape[0], a.shape[1] * b.shape[1], dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().curv_kron(_lib.stream_ptr(), a.data_ptr(), a.shape[0], a.shape[1], b.data_ptr(),
                                    b.shape[0], b.shape[1], out.data_ptr()), "curv_kron")
    return out


def concat(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """1-D concatenation of contiguous float32 GPU tensors: one batched copy launch."""
    for t in parts:
        _require_gpu(t)
    total = sum(t.numel() for t in parts)
    out = torch.empty(total, dtype=torch.float32, device=parts[0].device)
    pos, dsts = 0, []
    for t in parts:
        dsts.append(out[pos:pos + t.numel()])
        pos += t.numel()
    CopyPlan(dsts, [t.reshape(-1) for t in parts]).run()
    return out


TRI64_A_LOWER, TRI64_A_UPPER, TRI64_B_LOWER, TRI64_B_UPPER, TRI64_C_LOWER = 1, 2, 4, 8, 16      # CURV_TRI64_*


class Gemm64:
    """float64 C = alpha * A @ B [+ beta * C] on strided 2-D GPU views; C = None allocates the output.
    `tri`: TRI64_* flags for triangular operands (their other triangle must hold zeros; only the K range that can
    contribute to a tile is visited); TRI64_C_LOWER for a square product known to be symmetric (tiles strictly above the
    diagonal of C are not computed and keep what they held)."""
    __slots__ = ("A", "B", "C", "alpha", "beta", "tri", "E", "row_scale", "col_scale", "out32")

    def __init__(self, A, B, C=None, alpha=1.0, beta=0.0, tri=0, E=None, row_scale=None, col_scale=None, out32=None):
        """`E` (float64, shape of C, not C itself): C = alpha A B + beta E.  `out32` (float32, shape of the product): the
        result goes there instead, as float32(alpha * row_scale[i] * col_scale[j] * (A B)[i, j]) - either scale vector
        (float32, length M / N) may be None; no C, beta or E in that form."""
        self.A, self.B, self.C, self.alpha, self.beta, self.tri = A, B, C, float(alpha), float(beta), int(tri)
        self.E, self.row_scale, self.col_scale, self.out32 = E, row_scale, col_scale, out32


def gemm_f64_batched(jobs: Sequence[Gemm64]) -> List[torch.Tensor]:
    """All products through one call of curv_gemm_f64_batched (up to 24 descriptors per launch; the products must be
    independent of each other: large ones run in a launch of their own behind the small ones)."""
    n = len(jobs)
    if n == 0:
        return []
    d = (curv_gemm64_desc * n)()
    outs = []
    for k, j in enumerate(jobs):
        for t in (j.A, j.B):
            if not t.is_cuda or t.dtype != torch.float64 or t.dim() != 2:
                raise RuntimeError("gemm_f64 operands must be 2-D float64 GPU tensors")
        M, K = j.A.shape
        K2, N = j.B.shape
        if K != K2:
            raise RuntimeError("gemm_f64: shape mismatch")
        C = j.C
        if j.out32 is not None:
            o = j.out32
            if not o.is_cuda or o.dtype != torch.float32 or tuple(o.shape) != (M, N) or j.beta != 0.0 or j.E is not None \
                    or C is not None:
                raise RuntimeError("gemm_f64: the float32 output is an (M, N) float32 GPU tensor and takes no C / beta / E")
            for v, length in ((j.row_scale, M), (j.col_scale, N)):
                if v is not None and (not v.is_cuda or v.dtype != torch.float32 or not v.is_contiguous() or v.numel() != length):
                    raise RuntimeError("gemm_f64: scale vectors are contiguous float32 GPU tensors of length M / N")
            outs.append(o)
            d[k].A, d[k].B, d[k].C, d[k].C32 = j.A.data_ptr(), j.B.data_ptr(), None, o.data_ptr()
            d[k].row_scale = j.row_scale.data_ptr() if j.row_scale is not None else None
            d[k].col_scale = j.col_scale.data_ptr() if j.col_scale is not None else None
            d[k].c_rs, d[k].c_cs = o.stride()
        else:
            if C is None:
                if j.beta != 0.0 and j.E is None:
                    raise RuntimeError("gemm_f64: beta needs an existing C (or E)")
                C = torch.empty(M, N, dtype=torch.float64, device=j.A.device)
            elif not C.is_cuda or C.dtype != torch.float64 or tuple(C.shape) != (M, N):
                raise RuntimeError("gemm_f64: bad output tensor")
            if j.E is not None:
                E = j.E
                if not E.is_cuda or E.dtype != torch.float64 or tuple(E.shape) != (M, N) or E.stride() != C.stride() \
                        or E.data_ptr() == C.data_ptr():
                    raise RuntimeError("gemm_f64: E is a float64 GPU tensor with C's shape and strides, not C itself")
                d[k].E = E.data_ptr()
            outs.append(C)
            d[k].A, d[k].B, d[k].C = j.A.data_ptr(), j.B.data_ptr(), C.data_ptr()
            d[k].c_rs, d[k].c_cs = C.stride()
        d[k].a_rs, d[k].a_cs = j.A.stride()
        d[k].b_rs, d[k].b_cs = j.B.stride()
        d[k].M, d[k].N, d[k].K, d[k].alpha, d[k].beta, d[k].tri = M, N, K, j.alpha, j.beta, j.tri
    _lib.check(_lib.lib().curv_gemm_f64_batched(_lib.stream_ptr(), d, n), "curv_gemm_f64_batched")
    return outs


def gemm_f64(A: torch.Tensor, B: torch.Tensor, alpha: float = 1.0, beta: float = 0.0,
             C: Optional[torch.Tensor] = None) -> torch.Tensor:
    """float64 C = alpha * A @ B [+ beta * C] on strided 2-D GPU views."""
    return gemm_f64_batched([Gemm64(A, B, C, alpha, beta)])[0]
