#!/usr/bin/env python
"""Is the factor build power / clock limited?  Per-update time of the ResNet-50 factor build (tools/bench_syrk.py's jobs)
back to back versus with idle gaps between the calls, with rocm-smi's clock / power readings sampled beside both."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from curvature_amd import models, ops  # noqa: E402
from bench_syrk import make_jobs  # noqa: E402


def smi_sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5)
            out.append((time.time(), r.stdout.strip().replace("\n", " | ")))
        except Exception as exc:  # noqa: BLE001
            out.append((time.time(), f"smi failed: {exc}"))
        time.sleep(0.25)


def run(jobs, n, gap):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        ops.kfac_accumulate(jobs)
        b.record()
        if gap > 0:
            torch.cuda.synchronize()
            time.sleep(gap)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0], ts[-1]


def main():
    dev = torch.device("cuda:0")
    jobs, _ = make_jobs(models.resnet50(), (3, 224, 224), 32, dev)
    for _ in range(3):
        ops.kfac_accumulate(jobs)
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=smi_sampler, args=(stop, samples))
    th.start()
    for label, n, gap in (("back to back", 600, 0.0), ("20 ms gaps", 150, 0.02), ("back to back", 600, 0.0), ("100 ms gaps", 40, 0.1)):
        t0 = time.time()
        med, lo, hi = run(jobs, n, gap)
        print(f"{label:14s} n={n}: median {med:.3f} ms  min {lo:.3f}  max {hi:.3f}   [{t0:.2f} .. {time.time():.2f}]", flush=True)
    stop.set()
    th.join()
    for t, s in samples[:: max(1, len(samples) // 40)]:
        print(f"{t:.2f} {s[:300]}")


if __name__ == "__main__":
    main()
