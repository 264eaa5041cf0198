#!/usr/bin/env python
"""update() alone on the headline workload (ResNet-50, N = 32), for builds whose factors are wrong by construction
(tools/make_flat_ablate.py) and cannot go through invert():   CURV_ALT_LIB=tools/micro/libcurv_flat_ab4.so python tools/ab_update.py
Prints the median of 30 update() calls (HIP events on the caller's stream) and the clock / power rocm-smi reports meanwhile."""
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvature_amd import _lib  # noqa: E402

if os.environ.get("CURV_ALT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CURV_ALT_LIB"])
import torch  # noqa: E402
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = models.resnet50().to(dev).train()
kfac = KFAC(model)
x = torch.randn(32, 3, 224, 224, device=dev)
logits = model(x)
labels = torch.distributions.Categorical(logits=logits.detach()).sample()
torch.nn.functional.cross_entropy(logits, labels).backward()
for _ in range(5):
    kfac.update(32)
torch.cuda.synchronize()
smi, stop = [], threading.Event()


def sample():
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5)
            rows = [l for l in r.stdout.strip().splitlines() if l]
            if len(rows) >= 2:
                smi.append(dict(zip(rows[0].split(","), rows[1].split(","))))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.2)


th = threading.Thread(target=sample, daemon=True)
th.start()
ts = []
t_end = time.time() + 2.5
while time.time() < t_end or len(ts) < 30:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    kfac.update(32)
    e1.record()
    e1.synchronize()
    ts.append(e0.elapsed_time(e1))
stop.set()
th.join()


def col(part):
    vals = []
    for s in smi[1:]:
        for k, v in s.items():
            if part in k.lower():
                try:
                    vals.append(float(v.strip().strip("()").replace("Mhz", "")))
                except ValueError:
                    pass
                break
    return statistics.median(vals) if vals else float("nan")


print(f"{os.environ.get('CURV_ALT_LIB', 'product')}: update {statistics.median(ts):.3f} ms (min {min(ts):.3f}, {len(ts)} calls), "
      f"power {col('power'):.0f} W, sclk {col('sclk'):.0f} MHz")
