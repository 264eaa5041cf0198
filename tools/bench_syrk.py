#!/usr/bin/env python
"""Kernel microbenchmark: KFAC factor build (grouped SYRK) on the layer shapes of a model.

    python tools/bench_syrk.py --model resnet50 --batch 32 [--per-layer]

Inputs follow SURVEY.md section 8(d): x = max(0, N(0,1)), g = N(0,1)/N, fp32, seed = layer index.
Reports executed (symmetric) and dense-equivalent TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, ops  # noqa: E402

PEAK_F32_MFMA = 157.3e12


def conv_layers(model, chw):
    """(module, input shape, output shape) per selected layer, in modules() order."""
    shapes = {}
    hs = []
    for m in model.modules():
        if m.__class__.__name__ in ("Conv2d", "Linear"):
            hs.append(m.register_forward_hook(
                lambda mod, i, o: shapes.__setitem__(mod, (tuple(i[0].shape), tuple(o.shape)))))
    model.eval()
    with torch.no_grad():
        model(torch.zeros(1, *chw))
    for h in hs:
        h.remove()
    return [(m, *shapes[m]) for m in model.modules() if m in shapes]


def make_jobs(model, chw, N, dev):
    jobs, meta = [], []
    for idx, (m, ishape, oshape) in enumerate(conv_layers(model, chw)):
        torch.manual_seed(idx)
        x = torch.relu(torch.randn(N, *ishape[1:])).to(dev)
        g = (torch.randn(N, *oshape[1:]) / N).to(dev)
        bias = m.bias is not None
        if m.__class__.__name__ == "Conv2d":
            k, s, p = m.kernel_size, m.stride, m.padding
            L = oshape[2] * oshape[3]
            n = m.in_channels * k[0] * k[1] + int(bias)
        else:
            k, s, p, L = (1, 1), (1, 1), (0, 0), 1
            n = m.in_features + int(bias)
        mm = oshape[1]
        A = torch.zeros(n, n, device=dev)
        G = torch.zeros(mm, mm, device=dev)
        jobs.append(ops.FactorJob(x, A, k, s, p, bias, 1.0 / (N * L), False))
        jobs.append(ops.FactorJob(g, G, (1, 1), (1, 1), (0, 0), False, N / L, False))
        K = N * L
        meta.append((idx, n, mm, K))
    return jobs, meta


def time_jobs(jobs, iters, warmup=2):
    for _ in range(warmup):
        ops.kfac_accumulate(jobs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.kfac_accumulate(jobs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--per-layer", action="store_true")
    ap.add_argument("--classes", action="store_true", help="time factor classes (grouped per class)")
    ap.add_argument("--only", default="", help="restrict to one class, e.g. 3x3s1:2304 or G:1024")
    ap.add_argument("--marginal", action="store_true",
                    help="per class: time of the full launch minus the launch without that class (what the class costs "
                         "inside the grouped launch; isolated timings of small classes are dominated by launch effects)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    model, chw = {"lenet5": (models.lenet5, (1, 28, 28)), "resnet18": (models.resnet18, (3, 224, 224)),
                  "resnet50": (models.resnet50, (3, 224, 224))}[args.model]
    model = model()
    jobs, meta = make_jobs(model, chw, args.batch, dev)
    if args.only:
        kind_w, dim_w = args.only.split(":")
        sel, ex = [], 0.0
        for li, (idx, n, m, K) in enumerate(meta):
            for side, job, d in (("A", jobs[2 * li], n), ("G", jobs[2 * li + 1], m)):
                k = job.kernel
                kind = f"{k[0]}x{k[1]}s{job.stride[0]}" if side == "A" else "G"
                if kind == kind_w and d == int(dim_w):
                    sel.append(job)
                    ex += d * (d + 1.0) * K
        t = time_jobs(sel, args.iters)
        print(f"{args.only} x{len(sel)}: {t * 1e6:.1f} us  {ex / t / 1e12:.1f} TF/s exec ({ex / t / PEAK_F32_MFMA * 100:.1f}%)")
        return
    dense = sum(2.0 * (n * n + m * m) * K for _, n, m, K in meta)
    execd = sum(1.0 * (n * (n + 1) + m * (m + 1)) * K for _, n, m, K in meta)
    t = time_jobs(jobs, args.iters)
    print(f"{args.model} N={args.batch}: {t * 1e3:.3f} ms/update  executed {execd / t / 1e12:.1f} TFLOP/s "
          f"({execd / t / PEAK_F32_MFMA * 100:.1f}% of fp32 MFMA peak)  dense-equivalent {dense / t / 1e12:.1f} TFLOP/s")
    if args.marginal:
        classes = {}
        for li, (idx, n, m, K) in enumerate(meta):
            for side, job, d in (("A", jobs[2 * li], n), ("G", jobs[2 * li + 1], m)):
                k = job.kernel
                kind = f"{k[0]}x{k[1]}s{job.stride[0]}" if side == "A" else "G"
                classes.setdefault((kind, d, K), []).append(job)
        full = min(time_jobs(jobs, 5, 2) for _ in range(3))
        print(f"  full launch {full * 1e3:.3f} ms")
        rows = []
        for key, js in classes.items():
            ids = {id(j) for j in js}
            rest = [j for j in jobs if id(j) not in ids]
            t = min(time_jobs(rest, 5, 1) for _ in range(3))
            kind, d, K = key
            ex = d * (d + 1.0) * K * len(js)
            rows.append((full - t, kind, d, K, len(js), ex))
        for dt, kind, d, K, cnt, ex in sorted(rows, reverse=True):
            tf = ex / dt / 1e12 if dt > 0 else float("inf")
            print(f"  {kind:6s} dim={d:5d} K={K:7d} x{cnt:2d}: marginal {dt * 1e6:8.1f} us ({dt / full * 100:5.1f}% of the launch)  "
                  f"{tf:7.1f} TF/s exec  share of flops {ex / execd * 100:4.1f}%")
        print(f"  sum of marginals {sum(r[0] for r in rows) * 1e3:.3f} ms")
    if args.classes:
        classes = {}
        for li, (idx, n, m, K) in enumerate(meta):
            for side, job, d in (("A", jobs[2 * li], n), ("G", jobs[2 * li + 1], m)):
                k = job.kernel
                kind = f"{k[0]}x{k[1]}s{job.stride[0]}" if side == "A" else "G"
                key = (kind, d, K)
                classes.setdefault(key, []).append(job)
        for (kind, d, K), js in sorted(classes.items(), key=lambda kv: -kv[0][1] * kv[0][1] * kv[0][2] * len(kv[1])):
            tt = time_jobs(js, 3, 1)
            ex = d * (d + 1.0) * K * len(js)
            print(f"  {kind:6s} dim={d:5d} K={K:7d} x{len(js):2d}: {tt * 1e6:9.1f} us  {ex / tt / 1e12:6.1f} TF/s exec "
                  f"({ex / tt / PEAK_F32_MFMA * 100:5.1f}%)  share of flops {ex / execd * 100:4.1f}%")
    if args.per_layer:
        for li, (idx, n, m, K) in enumerate(meta):
            for side, job, d in (("A", jobs[2 * li], n), ("G", jobs[2 * li + 1], m)):
                tt = time_jobs([job], 3, 1)
                ex = d * (d + 1.0) * K
                print(f"  layer {idx:2d} {side} dim={d:5d} K={K:7d}: {tt * 1e6:9.1f} us  "
                      f"{ex / tt / 1e12:6.1f} TF/s exec ({ex / tt / PEAK_F32_MFMA * 100:5.1f}%)")


if __name__ == "__main__":
    main()
