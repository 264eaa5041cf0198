#!/usr/bin/env python
"""Where the eigensolver's time goes on the 108 ResNet-50 factors (N = 32) with the low-rank path: the path alone on the
factors it takes, the iteration on the rest, one 4608-wide factor alone either way."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models, ops
from curvature_amd.curvatures import KFAC
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = models.resnet50().to(dev).train()
kfac = KFAC(model)
x = torch.randn(32, 3, 224, 224, device=dev)
logits = model(x)
labels = torch.distributions.Categorical(logits=logits.detach()).sample()
torch.nn.functional.cross_entropy(logits, labels).backward()
kfac.update(32)
mats = [f for l in kfac._layers() for f in kfac.state[l]]


def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


big = [m for m in mats if m.shape[0] == 4608]
print("sizes >= 2048:", sorted((m.shape[0] for m in mats if m.shape[0] >= 2048), reverse=True))
print(f"all 108: {timed(lambda: ops.eigh(mats)):.0f} ms, low-rank path took {ops.eigh.last_lowrank}")
print(f"three 4608: {timed(lambda: ops.eigh(big)):.0f} ms (low-rank {ops.eigh.last_lowrank})")
print(f"one 4608: {timed(lambda: ops.eigh(big[:1])):.0f} ms")
rest = [m for m in mats if m.shape[0] != 4608]
print(f"the other 105: {timed(lambda: ops.eigh(rest)):.0f} ms (low-rank {ops.eigh.last_lowrank})")
os.environ["CURV_EIGH_LOWRANK"] = "0"
print(f"one 4608, iteration on the whole matrix: {timed(lambda: ops.eigh(big[:1])):.0f} ms")
print(f"all 108, iteration on the whole matrices: {timed(lambda: ops.eigh(mats)):.0f} ms")
