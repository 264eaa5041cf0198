#!/usr/bin/env python
"""Whole-model invert(1, 1000) of the 108 ResNet-50 factors with k unrelated streams created BEFORE the library's own
stream set (what RCCL, a data loader or eval_bnn(overlap=True) add to a process).  One process per k (the set is created
once per process); env switches of the library are passed through.
    python tools/stream_sensitivity.py            # k = 0 .. 5, child processes
    python tools/stream_sensitivity.py --child K  # one measurement"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(k):
    import torch
    from curvature_amd import models, ops
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    if os.environ.get("SET_FIRST"):                            # the library's stream set exists before the unrelated streams
        A = torch.eye(64, device=dev) * 2
        ops.chol_inv_lower([A], [1.0], [1.0], check=True)
        torch.cuda.synchronize()
    if os.environ.get("DUMMY_KIND", "torch") == "raw":         # hipStreamCreateWithFlags(nonblocking), as RCCL / MIOpen do
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        dummies = [ctypes.c_void_p() for _ in range(k)]
        for d in dummies:
            assert hip.hipStreamCreateWithFlags(ctypes.byref(d), 1) == 0
    else:
        dummies = [torch.cuda.Stream() for _ in range(k)]      # noqa: F841  (kept alive, never used)
    if os.environ.get("WITH_MODEL"):                           # a forward / backward pass first (MIOpen, rocBLAS handles)
        m = models.resnet50().to(dev).train()
        x = torch.randn(8, 3, 224, 224, device=dev)
        m(x).sum().backward()
        torch.cuda.synchronize()
    rows = models.layer_table(models.resnet50(), (3, 224, 224))
    sizes = [d for r in rows for d in (r["n"], r["m"])]
    Fs = []
    for i, n in enumerate(sizes):
        torch.manual_seed(i)
        kk = min(n + 8, 4096)
        X = torch.randn(n, kk, device=dev)
        Fs.append((X @ X.t() / kk).contiguous())
    add, mul = [1.0] * len(Fs), [1000.0] * len(Fs)
    outs = None
    for _ in range(3):
        outs = ops.chol_inv_lower(Fs, add, mul, check=True, outs=outs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        outs = ops.chol_inv_lower(Fs, add, mul, check=True, outs=outs)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    import hashlib
    h = hashlib.sha256()
    for t in outs:
        h.update(t.cpu().numpy().tobytes())
    print(f"k={k} invert median {ts[len(ts) // 2]:.2f} ms  min {ts[0]:.2f}  sha {h.hexdigest()[:12]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        ks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
        for k in ks:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(k)], check=False)
