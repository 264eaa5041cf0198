#!/usr/bin/env python
"""update() alone on a model (resnet50 / resnet18 / densenet121), median of back-to-back calls; for A/B of library builds
(CURV_ALT_LIB) and kernel statistics under rocprofv3:   python tools/update_only.py densenet121 [iters]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curvature_amd import _lib  # noqa: E402

if os.environ.get("CURV_ALT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CURV_ALT_LIB"])
import torch  # noqa: E402
from curvature_amd import models  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = getattr(models, name)().to(dev).train()
kfac = KFAC(model)
x = torch.randn(32, 3, 224, 224, device=dev)
logits = model(x)
labels = torch.distributions.Categorical(logits=logits.detach()).sample()
torch.nn.functional.cross_entropy(logits, labels).backward()
for _ in range(5):
    kfac.update(32)
torch.cuda.synchronize()
ts = []
for _ in range(iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    kfac.update(32)
    e1.record()
    e1.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"{name}: update {statistics.median(ts):.3f} ms (min {min(ts):.3f}, {iters} calls)")
