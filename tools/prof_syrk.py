#!/usr/bin/env python
"""Phase-cycle breakdown of syrk_patch_kernel from an instrumented build (tools/micro/libcurv_prof.so,
built from syrk.hip + clock64() probes; diagnostics only, never shipped)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libcurv_prof.so")
from curvature_amd import models, ops  # noqa: E402
import bench_syrk  # noqa: E402

NAMES = ["prologue", "ktab", "store_stage", "sync1", "decode+issue_loads", "mfma_loop", "sync2", "epilogue"]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    dev = torch.device("cuda:0")
    model = models.resnet50()
    jobs, meta = bench_syrk.make_jobs(model, (3, 224, 224), 32, dev)
    if only:
        kind, dim = only.split(":")     # e.g. 3x3:2304, 1x1:1024
        k = int(kind[0])
        jobs = [j for j in jobs if j.dst.shape[0] == int(dim) and j.kernel[0] == k]
    h = _lib.lib()
    h.curv_debug_syrk_prof.restype = ctypes.c_int
    h.curv_debug_syrk_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    ops.kfac_accumulate(jobs)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    h.curv_debug_syrk_prof(buf, 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ops.kfac_accumulate(jobs)
    ev1.record()
    torch.cuda.synchronize()
    h.curv_debug_syrk_prof(buf, 0)
    v = list(buf)
    tot = sum(v[:8]) + v[8]
    print(f"launch {ev0.elapsed_time(ev1):.3f} ms; wave-cycles total {tot:.3e}; items(waves) {v[11]}; MFMA {v[10]}")
    for n, c in zip(NAMES, v[:8]):
        print(f"  {n:20s} {c:.3e}  {100.0 * c / tot:5.1f}%")
    print(f"  wait for staged loads (vmcnt) before the store phase: {v[8]:.3e}  {100.0 * v[8] / tot:5.1f}%")
    nch = max(v[12], 1)
    print(f"  wave-chunks {v[12]}; per wave-chunk cycles: " + ", ".join(f"{n}={c / nch:.0f}" for n, c in zip(NAMES, v[:8])))
    print(f"  cycles per MFMA inside the loop: {v[5] / max(v[10], 1):.1f}  (64 = pipe-bound for one wave, "
          f"128 = two waves sharing a SIMD)")
    timeline(h)


def timeline(h):
    import numpy as np
    buf = (ctypes.c_ulonglong * (3 * 16384))()
    h.curv_debug_syrk_times.restype = ctypes.c_int
    h.curv_debug_syrk_times.argtypes = [ctypes.c_void_p]
    h.curv_debug_syrk_times(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 3)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0).astype(np.float64) / 100.0, (a[:, 1] - t0).astype(np.float64) / 100.0   # us (100 MHz)
    span = en.max()
    print(f"  timeline: {len(a)} items, span {span:.0f} us, sum of item time {np.sum(en - st) / 512:.0f} us per slot "
          f"(occupancy {np.sum(en - st) / 512 / span:.2f})")
    for frac in (0.5, 0.8, 0.9, 0.95, 1.0):
        t = span * frac
        print(f"    active items at {frac:.2f} span: {int(np.sum((st <= t) & (en > t)))}")
    keys = {}
    for (s0, e0, k) in zip(st, en, a[:, 2]):
        keys.setdefault(int(k), []).append(e0 - s0)
    print("    item duration by (dim, kh, TM):")
    for k in sorted(keys, key=lambda k: -np.sum(keys[k])):
        v = np.array(keys[k])
        print(f"      dim {k >> 32:5d} k{(k & 0xffffffff) // 100} TM{(k & 0xffffffff) % 100:3d}: n={len(v):5d} mean {v.mean():7.1f} us  "
              f"max {v.max():7.1f}  total/512 {v.sum() / 512:7.1f} us")


if __name__ == "__main__":
    main()
