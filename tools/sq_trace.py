#!/usr/bin/env python
"""Time stamps inside chol_square_kernel (csrc/invert.hip) for one panel of one 4608-wide factor: builds the library
with -DCURV_SQ_TRACE=<panel + 1> into tools/micro/libcurv_sqtrace.so (`--build-only` here, then run on the GPU box).
Diagnostics only."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "micro", "libcurv_sqtrace.so")
PANEL = 9


def build():
    from curvature_amd import _lib
    srcs = [os.path.join(_lib.CSRC, f) for f in _lib.SOURCES]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", f"-DCURV_SQ_TRACE={PANEL}",
           "-I" + os.path.join(ROOT, "include"), "-o", LIB] + srcs
    subprocess.check_call(cmd)


def main():
    if "--build-only" in sys.argv:
        build()
        return
    from curvature_amd import _lib
    _lib.LIB_PATH = LIB
    import torch
    from curvature_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
    X = torch.randn(n, 4096, device=dev)
    F = (X @ X.t() / 4096).contiguous()
    lib = ctypes.CDLL(LIB)
    for _ in range(3):
        ops.chol_inv_lower([F], [1.0], [1000.0], check=False)
    torch.cuda.synchronize()
    assert lib.curv_debug_sq_trace_reset() == 0
    ops.chol_inv_lower([F], [1.0], [1000.0], check=False)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 256)()
    assert lib.curv_debug_sq_trace(out) == 0
    t = list(out)
    t0 = t[0]
    us = lambda v: (v - t0) / 100.0 if v else float("nan")
    print("runner: start 0 | loaded %.1f | D0 done %.1f | FD0 posted %.1f" % (us(t[1]), us(t[2]), us(t[3])))
    for q in (1, 2, 3):
        b = 8 * q
        print("  step %d: FT seen %.1f | operands in LDS %.1f | solve %.1f | A' ready %.1f | FLC posted %.1f | D done %.1f | FD posted %.1f"
              % (q, us(t[b]), us(t[b + 1]), us(t[b + 2]), us(t[b + 3]), us(t[b + 4]), us(t[b + 5]), us(t[b + 6])))
    print("  inside the factorisations (us after their start): " + " | ".join(
        "step %d: " % q + " ".join("%.1f" % ((t[112 + 4 * q + p] - (t[1] if q == 0 else t[8 * q + 4])) / 100.0) for p in range(4)) for q in range(4)))
    for r in (1, 2, 3):
        b = 64 * r
        ms = " ".join("m%d: T %.1f X seen %.1f L %.1f |" % (m, us(t[b + 8 + 4 * m]), us(t[b + 9 + 4 * m]), us(t[b + 10 + 4 * m])) for m in range(max(r - 1, 0)))
        cols = " ".join("X[%d][%d]: S done %.1f, X_ii seen %.1f, stored %.1f |" % (i, r - 1, us(t[b + 2 + 16 * (i - r)]), us(t[b + 3 + 16 * (i - r)]), us(t[b + 4 + 16 * (i - r)])) for i in range(r, 4))
        print("helper %d: start %.1f | %s FT posted %.1f | %s" % (r, us(t[b]), ms, us(t[b + 1]), cols))
    p0 = t[240]
    print("panel product (block 0): body start 0 | own operand issued, first tiles requested %.1f | after tile k: %s"
          % ((t[241] - p0) / 100.0, " ".join("%.1f" % ((t[242 + k] - p0) / 100.0) for k in range(10))))
    big = 1 << 62
    kt = lambda s_, e_: ((big - t[s_] - t0) / 100.0 if t[s_] else float("nan"), (t[e_] - t0) / 100.0 if t[e_] else float("nan"))
    print("launches of this panel, first workgroup in -> last workgroup out (us after the square kernel's start): "
          "panel product %.1f -> %.1f | near update %.1f -> %.1f | far update %.1f -> %.1f | next panel's square kernel starts %.1f"
          % (kt(236, 237) + kt(252, 253) + kt(254, 255) + (us(t[238]),)))


if __name__ == "__main__":
    main()
