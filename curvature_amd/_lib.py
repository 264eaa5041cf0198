"""ctypes binding of ``libcurv_hip.so`` (the C ABI declared in ``include/curv_hip.h``).

The shared library is built in-tree by :func:`build` (``hipcc --offload-arch=gfx950``) and loaded
lazily by :func:`lib`.  There is no CPU fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised.
"""
import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(_HERE, "..", "include"))
LIB_PATH = os.path.join(CSRC, "libcurv_hip.so")
SOURCES = ["api.cpp", "collective.cpp", "elementwise.hip", "syrk.hip", "syrk_flat.hip", "syrk_corr.hip", "syrk_pre.hip", "syrk_small.hip", "invert.hip", "gemm.hip", "inf.hip", "eigh.hip", "eigh_lowrank.hip"]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", "-ldl"]

_lock = threading.Lock()
_lib = None


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    deps.append(os.path.join(INCLUDE, "curv_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _compiler_id(hipcc: str) -> str:
    try:
        return subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    except OSError as exc:
        raise RuntimeError(f"cannot run {hipcc}: {exc}")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``csrc/libcurv_hip.so``; returns the library path.  Each source is
    compiled to an object of its own under ``csrc/build/`` (in parallel), then linked.  An object is reused only while
    the stamp beside it - a hash over compiler version, flags, the source and every header - still matches (an mtime
    comparison would link stale objects after a change of flags, ``$HIPCC`` or ROCm).  Objects and the library are
    written to temporary names and moved into place, under a file lock on ``csrc/build``: several ranks of a torchrun
    job may call this at once on a fresh checkout."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "curv_hip.h")])
    compile_flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-ldl")] + ["-c"]
    base = hashlib.sha256()
    base.update(_compiler_id(hipcc).encode())
    base.update(" ".join(compile_flags).encode())
    for h in headers:
        base.update(open(h, "rb").read())

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        key = base.copy()
        key.update(open(path, "rb").read())
        stamp, want = obj + ".stamp", key.hexdigest()
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
            return obj, 0, ""
        tmp = f"{obj}.{os.getpid()}.tmp"
        cmd = [hipcc] + compile_flags + ["-o", tmp, path]
        if verbose:
            print(" ".join(cmd[:-3] + ["-o", obj, path]), flush=True)
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode == 0:
            os.replace(tmp, obj)
            with open(stamp + ".tmp", "w") as fh:
                fh.write(want)
            os.replace(stamp + ".tmp", stamp)
        elif os.path.exists(tmp):
            os.remove(tmp)
        return obj, proc.returncode, proc.stdout

    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():                 # another process built it while this one waited
                return LIB_PATH
            with ThreadPoolExecutor(max_workers=int(os.environ.get("CURV_BUILD_JOBS", "6"))) as pool:
                results = list(pool.map(compile_one, SOURCES))
            for obj, rc, out in results:
                if rc != 0:
                    raise RuntimeError("hipcc failed:\n" + out)
                if verbose and out:
                    print(out)
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + [r[0] for r in results] + ["-ldl"]
            if verbose:
                print(" ".join(cmd), flush=True)
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if proc.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise RuntimeError("hipcc (link) failed:\n" + proc.stdout)
            os.replace(tmp, LIB_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


class curv_factor_desc(ctypes.Structure):
    """Mirror of ``curv_factor_desc`` in include/curv_hip.h."""
    _fields_ = [
        ("src", ctypes.c_void_p), ("dst", ctypes.c_void_p),
        ("N", ctypes.c_int32), ("C", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("sh", ctypes.c_int32), ("sw", ctypes.c_int32),
        ("ph", ctypes.c_int32), ("pw", ctypes.c_int32),
        ("has_bias", ctypes.c_int32), ("first", ctypes.c_int32),
        ("scale", ctypes.c_float), ("path_hint", ctypes.c_int32),
    ]


class curv_inv_desc(ctypes.Structure):
    """Mirror of ``curv_inv_desc`` in include/curv_hip.h."""
    _fields_ = [("F", ctypes.c_void_p), ("L", ctypes.c_void_p), ("n", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("add", ctypes.c_double), ("multiply", ctypes.c_double)]


class curv_gemm_desc(ctypes.Structure):
    """Mirror of ``curv_gemm_desc`` in include/curv_hip.h."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("E", ctypes.c_void_p)] + \
               [(k, ctypes.c_longlong) for k in ("a_rs", "a_cs", "b_rs", "b_cs", "c_rs", "c_cs", "e_rs", "e_cs")] + \
               [("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("epilogue", ctypes.c_int32),
                ("alpha", ctypes.c_float), ("beta", ctypes.c_float), ("tri", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("F", ctypes.c_void_p), ("f_rs", ctypes.c_longlong),
                ("f_cs", ctypes.c_longlong)]


class curv_cholinv_desc(ctypes.Structure):
    _fields_ = [("M", ctypes.c_void_p), ("X", ctypes.c_void_p), ("n", ctypes.c_int32), ("m_is_f64", ctypes.c_int32),
                ("diag_add", ctypes.c_double), ("R", ctypes.c_void_p), ("r_minus", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("pivot_min", ctypes.c_double)]


class curv_gemm64_desc(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p)] + \
               [(k, ctypes.c_longlong) for k in ("a_rs", "a_cs", "b_rs", "b_cs", "c_rs", "c_cs")] + \
               [("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("tri", ctypes.c_int32),
                ("alpha", ctypes.c_double), ("beta", ctypes.c_double),
                ("E", ctypes.c_void_p), ("row_scale", ctypes.c_void_p), ("col_scale", ctypes.c_void_p), ("C32", ctypes.c_void_p)]


class curv_sq_desc(ctypes.Structure):
    _fields_ = [("grad_w", ctypes.c_void_p), ("grad_b", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("rows", ctypes.c_int32), ("cols_w", ctypes.c_int32), ("first", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class curv_copy_desc(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("src", ctypes.c_void_p), ("bytes", ctypes.c_ulonglong)]


class curv_eigh_desc(ctypes.Structure):
    _fields_ = [("F", ctypes.c_void_p), ("U", ctypes.c_void_p), ("w", ctypes.c_void_p), ("n", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class curv_select_desc(ctypes.Structure):
    _fields_ = [("lambda_vec", ctypes.c_void_p), ("I", ctypes.c_void_p), ("J", ctypes.c_void_p),
                ("counts", ctypes.c_void_p), ("n", ctypes.c_int32), ("m", ctypes.c_int32), ("rank", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


_vp, _i, _ll, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_double, ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/curv_hip.h declares
SIGNATURES = {
    "curv_version": (_i, []),
    "curv_last_error": (ctypes.c_char_p, []),
    "curv_init_streams": (_i, []),
    "curv_allgather_weights": (_i, [_vp, _vp, _vp, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]),
    "curv_rccl_available": (_i, []),
    "curv_comm_unique_id": (_i, [_vp]),
    "curv_comm_init": (_i, [ctypes.POINTER(_vp), _i, _vp, _i]),
    "curv_comm_destroy": (_i, [_vp]),
    "curv_kfac_workspace_bytes": (_sz, [ctypes.POINTER(curv_factor_desc), _i]),
    "curv_kfac_plan_info": (_i, [ctypes.POINTER(curv_factor_desc), _i, ctypes.POINTER(ctypes.c_longlong)]),
    "curv_kfac_path_for": (_i, [ctypes.POINTER(curv_factor_desc), _i]),
    "curv_kfac_accumulate": (_i, [_vp, ctypes.POINTER(curv_factor_desc), _i, _vp, _sz]),
    "curv_kfac_accumulate_timed": (_i, [_vp, ctypes.POINTER(curv_factor_desc), _i, _vp, _sz, _vp, _vp]),
    "curv_kfac_accumulate_ex": (_i, [_vp, ctypes.POINTER(curv_factor_desc), _i, _vp, _sz, ctypes.c_uint, _vp, _vp]),
    "curv_event_create": (_vp, []),
    "curv_event_destroy": (None, [_vp]),
    "curv_event_elapsed_ms": (_i, [_vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    "curv_event_synchronize": (_i, [_vp]),
    "curv_chol_inv_workspace_bytes": (_sz, [ctypes.POINTER(curv_inv_desc), _i]),
    "curv_chol_inv_lower": (_i, [_vp, ctypes.POINTER(curv_inv_desc), _i, _vp, _vp, _sz]),
    "curv_chol_inv_lower_status": (_i, [_vp, ctypes.POINTER(curv_inv_desc), _i, _vp, _vp, _sz, _vp, _vp]),
    "curv_chol_factor_inverse_workspace_bytes": (_sz, [ctypes.POINTER(curv_cholinv_desc), _i]),
    "curv_chol_factor_inverse": (_i, [_vp, ctypes.POINTER(curv_cholinv_desc), _i, _vp, _vp, _sz]),
    "curv_gemm_f64_batched": (_i, [_vp, ctypes.POINTER(curv_gemm64_desc), _i]),
    "curv_syevd_workspace_bytes": (_sz, [ctypes.POINTER(curv_eigh_desc), _i]),
    "curv_syevd": (_i, [_vp, ctypes.POINTER(curv_eigh_desc), _i, _vp, _sz, _i, _d, ctypes.POINTER(ctypes.c_int)]),
    "curv_syevd_ex": (_i, [_vp, ctypes.POINTER(curv_eigh_desc), _i, _vp, _sz, _i, _d, ctypes.POINTER(ctypes.c_int),
                          ctypes.POINTER(ctypes.c_int)]),
    "curv_inf_select": (_i, [_vp, ctypes.POINTER(curv_select_desc), _i]),
    "curv_colpairs": (_i, [_vp, _vp, _i, _i, _ll, _vp]),
    "curv_inf_vtv_assemble": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "curv_inf_vtv_assemble_f64": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "curv_colpairs_f64": (_i, [_vp, _vp, _i, _i, _ll, _vp]),
    "curv_colpairs_sym_f64": (_i, [_vp, _vp, _i, _i, _ll, _vp]),
    "curv_inf_vtv_assemble_sym_f64": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "curv_square_f64": (_i, [_vp, _vp, _vp, _ll]),
    "curv_diag_scale": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _i]),
    "curv_gather2d": (_i, [_vp, _vp, _ll, _ll, _vp, _vp, _vp, _i, _i]),
    "curv_kron": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "curv_mul2d": (_i, [_vp, _vp, _ll, _ll, _vp, _ll, _ll, _vp, _i, _i]),
    "curv_gemm_workspace_bytes": (_sz, [_i]),
    "curv_gemm_workspace_bytes_for": (_sz, [ctypes.POINTER(curv_gemm_desc), _i]),
    "curv_gemm_batched": (_i, [_vp, ctypes.POINTER(curv_gemm_desc), _i, _vp, _sz]),
    "curv_gemm_batched_ex": (_i, [_vp, ctypes.POINTER(curv_gemm_desc), _i, _vp, _sz, ctypes.c_uint]),
    "curv_randn": (_i, [_vp, _vp, _ll, ctypes.c_ulonglong, ctypes.c_ulonglong]),
    "curv_randn_counter": (_i, [_vp, _vp, ctypes.c_longlong, ctypes.c_ulonglong, _vp]),
    "curv_rsqrt_affine": (_i, [_vp, _vp, _d, _d, _vp, _ll]),
    "curv_sq_accumulate": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _i]),
    "curv_sq_accumulate_batched": (_i, [_vp, ctypes.POINTER(curv_sq_desc), _i, _d]),
    "curv_clamp_min0": (_i, [_vp, _vp, _ll]),
    "curv_sqrt_scale": (_i, [_vp, _vp, _d, _vp, _ll]),
    "curv_mul": (_i, [_vp, _vp, _vp, _vp, _ll]),
    "curv_copy_batched": (_i, [_vp, ctypes.POINTER(curv_copy_desc), _i]),
}


ABI_VERSION = 10                     # CURV_ABI_VERSION of include/curv_hip.h
KFAC_TABLE_RESIDENT = 1             # CURV_KFAC_TABLE_RESIDENT
PATH_AUTO, PATH_SMALL, PATH_GROUPED = 0, 1, 2     # CURV_PATH_* (curv_factor_desc.path_hint)
SMALL_MAX_FLOP = 2.0e9              # CURV_SMALL_MAX_FLOP
GEMM_TABLE_RESIDENT = 1             # CURV_GEMM_TABLE_RESIDENT
ERR_NOT_PD, ERR_INVALID, ERR_WORKSPACE, ERR_HIP, ERR_NOT_CONVERGED = 1, 2, 3, 4, 5     # CURV_ERR_* of the header


def lib() -> ctypes.CDLL:
    """Load the HIP library (once).  torch must be imported first so that both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (binds libamdhip64.so.7 of the torch wheel before our DT_NEEDED resolves)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.curv_version() != ABI_VERSION:
            raise RuntimeError("libcurv_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(status: int, what: str = "") -> None:
    """Turn a non-zero status of the C ABI into ``RuntimeError`` (error convention)."""
    if status != 0:
        msg = lib().curv_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libcurv_hip {what} failed with status {status}: {msg}")


def init_streams(device) -> None:
    """Create the library's internal streams for `device` now (curv_init_streams): the mapping of streams onto hardware
    queues depends on creation order, and the inversion sweep is fastest when its streams exist before any unrelated
    ones.  Called by the estimator constructors; a no-op for CPU devices and after the first call."""
    import torch
    device = torch.device(device)
    if device.type != "cuda":
        return
    with torch.cuda.device(device):
        # torch's own stream of this device must have been USED before the set is created: measured with the set created
        # first thing after set_device (before any torch kernel), invert() of the ResNet-50 factors takes 13.2 ms instead
        # of 8.3 - the runtime gives hardware queues out lazily, in order of first use
        torch.zeros(1, device=device).add_(1)
        torch.cuda.current_stream(device).synchronize()
        check(lib().curv_init_streams(), "curv_init_streams")


def stream_ptr() -> int:
    """hipStream_t of torch's current stream, so our kernels are ordered after backward()."""
    import torch
    return int(torch.cuda.current_stream().cuda_stream)
