#!/usr/bin/env python
"""Where the waves of syrk_flat_kernel spend their cycles (instrumented build, tools/make_prof_flat.py)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libcurv_prof_flat.so")
from curvature_amd import models, ops  # noqa: E402
import bench_syrk  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    jobs, meta = bench_syrk.make_jobs(models.resnet50(), (3, 224, 224), 32, dev)
    h = _lib.lib()
    h.curv_debug_flat_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    h.curv_debug_flat_times.argtypes = [ctypes.c_void_p]
    for _ in range(2):
        ops.kfac_accumulate(jobs)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    h.curv_debug_flat_prof(buf, 1)
    ops.kfac_accumulate(jobs)
    torch.cuda.synchronize()
    h.curv_debug_flat_prof(buf, 0)
    pro, wait, bar, work, nst, nw = list(buf)[:6]
    tot = pro + wait + bar + work
    print(f"waves {nw}, stages {nst}; wave-cycles: prologue {100 * pro / tot:.1f}%  vmcnt wait {100 * wait / tot:.1f}%  "
          f"barrier {100 * bar / tot:.1f}%  MFMA/issue {100 * work / tot:.1f}%")
    print(f"per stage: wait {wait / nst:.0f}  barrier {bar / nst:.0f}  work {work / nst:.0f} cycles")
    tb = (ctypes.c_ulonglong * (3 * 32768))()
    h.curv_debug_flat_times(tb)
    a = np.frombuffer(tb, dtype=np.uint64).reshape(-1, 3)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0).astype(np.float64) / 100.0, (a[:, 1] - t0).astype(np.float64) / 100.0
    span = en.max()
    print(f"timeline: {len(a)} items, span {span:.0f} us, mean active workgroups {np.sum(en - st) / span:.0f}")
    print("  active workgroups at 5 % steps of the span: " +
          " ".join(str(int(np.sum((st <= span * f) & (en > span * f)))) for f in np.arange(0.05, 1.0, 0.05)))
    keys = {}
    for (s0, e0, k) in zip(st, en, a[:, 2]):
        keys.setdefault(int(k), []).append((s0, e0))
    print("  class (dim, K per sample): n, mean / max duration, first start, last start, last end  [us]")
    for k in sorted(keys, key=lambda k: -max(e for _, e in keys[k])):
        v = np.array(keys[k])
        dur = v[:, 1] - v[:, 0]
        print(f"    dim {k >> 32:5d} K {k & 0xffffffff:6d}: n={len(v):5d} mean {dur.mean():7.1f} max {dur.max():7.1f}  "
              f"start {v[:, 0].min():7.1f} .. {v[:, 0].max():7.1f}  last end {v[:, 1].max():7.1f}  total/512 {dur.sum() / 512:7.1f}")


if __name__ == "__main__":
    main()
