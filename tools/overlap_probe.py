#!/usr/bin/env python
"""What would overlapping the factor build with the inversion bring?  ResNet-50, N = 32: update() on one stream and an
inversion of (a copy of) the model's factors on another, enqueued together, against the two run one after the other.
The data are independent here (the inversion reads a copy taken beforehand) - a timing probe, not a product path."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from curvature_amd import _lib, models, ops  # noqa: E402
from curvature_amd.curvatures import KFAC  # noqa: E402

dev = torch.device("cuda:0")
_lib.init_streams(dev)
torch.manual_seed(0)
model = models.resnet50().to(dev).train()
kfac = KFAC(model)
x = torch.randn(32, 3, 224, 224, device=dev)
logits = model(x)
labels = torch.distributions.Categorical(logits=logits.detach()).sample()
torch.nn.functional.cross_entropy(logits, labels).backward()
kfac.update(32)
torch.cuda.synchronize()
Fs = [F.clone() for layer in kfac._layers() for F in kfac.state[layer]]
add, mul = [1.0] * len(Fs), [1000.0] * len(Fs)
big = [F for F in Fs if F.shape[0] > 2304]
small = [F for F in Fs if F.shape[0] <= 2304]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, iters=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


def seq():
    kfac.update(32)
    ops.chol_inv_lower(Fs, add, mul, check=False)


def par():
    with torch.cuda.stream(s2):
        ops.chol_inv_lower(Fs, add, mul, check=False)
    with torch.cuda.stream(s1):
        kfac.update(32)


def par_big():          # only the large factors' inversion beside the build, the rest behind it
    with torch.cuda.stream(s2):
        ops.chol_inv_lower(big, add[:len(big)], mul[:len(big)], check=False)
    with torch.cuda.stream(s1):
        kfac.update(32)
        ops.chol_inv_lower(small, add[:len(small)], mul[:len(small)], check=False)


print("update alone            %.3f ms" % timed(lambda: kfac.update(32)))
print("invert alone            %.3f ms" % timed(lambda: ops.chol_inv_lower(Fs, add, mul, check=False)))
print("invert big (3) alone    %.3f ms" % timed(lambda: ops.chol_inv_lower(big, add[:len(big)], mul[:len(big)], check=False)))
print("invert rest alone       %.3f ms" % timed(lambda: ops.chol_inv_lower(small, add[:len(small)], mul[:len(small)], check=False)))
print("update ; invert         %.3f ms" % timed(seq))
print("update || invert        %.3f ms" % timed(par))
print("update || big ; rest    %.3f ms" % timed(par_big))
