#!/usr/bin/env python
"""Power / clock readings (rocm-smi, 4 Hz) while each phase of the headline step runs back to back for ~3 s: which of
update() / invert() / sample_and_replace() sits at the board's regulation point (the factor build: a constant ~1.1 kW at a
reduced MFMA clock) and which does not."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402


def sampler(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5)
            out.append((time.time(), r.stdout))
        except Exception as exc:  # noqa: BLE001
            out.append((time.time(), f"smi failed: {exc}"))
        time.sleep(0.25)


def fields(text):
    rows = [l for l in text.strip().splitlines() if l]
    if len(rows) < 2:
        return text[:120]
    head, vals = rows[0].split(","), rows[1].split(",")
    keep = {}
    for h, v in zip(head, vals):
        hl = h.lower()
        if "power" in hl or "sclk" in hl or "mclk" in hl:
            keep[h.strip()] = v.strip()
    return keep


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = models.resnet50().to(dev)
    x = torch.randn(32, 3, 224, 224, device=dev)
    k = KFAC(model)
    logits = model(x)
    labels = torch.distributions.Categorical(logits=logits).sample()
    torch.nn.functional.cross_entropy(logits, labels).backward()
    k.update(32); k.invert(1.0, 1000.0); k.sample_and_replace()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, samples))
    th.start()
    marks = []
    for name, fn in (("idle", lambda: time.sleep(0.05)), ("update", lambda: (k.restart_accumulation(), k.update(32))), ("invert", lambda: k.invert(1.0, 1000.0)),
                     ("sample", k.sample_and_replace), ("idle", lambda: time.sleep(0.05))):
        t0 = time.time()
        n = 0
        while time.time() - t0 < 1.5:
            fn()
            n += 1
            if n % 20 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        marks.append((name, t0, time.time(), n))
        print("phase done", name, n, flush=True)
    stop.set()
    th.join()
    for name, t0, t1, n in marks:
        mine = [fields(s) for t, s in samples if t0 + 0.5 < t < t1 - 0.1]
        print(f"{name:7s} {n:6d} calls in {t1 - t0:.2f} s = {(t1 - t0) / n * 1e3:8.3f} ms per call; readings: {mine[:3]}")


if __name__ == "__main__":
    main()
