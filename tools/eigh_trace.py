import os, sys, torch
sys.path.insert(0, "/root/repo")
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
layers = kfac._layers()
names = {m: n for n, m in model.named_modules()}
sel = [l for l in layers if names[l] in ("layer4.1.conv2", "layer3.1.conv2", "layer2.1.conv1", "fc")]
mats = [f for l in sel for f in kfac.state[l]]
print([tuple(m.shape) for m in mats])
ops.eigh(mats)
print("sweeps", ops.eigh.last_sweeps)
import time
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vecs, vals = ops.eigh(mats, with_values=True)
    torch.cuda.synchronize(); print("time", time.perf_counter() - t0, "sweeps", ops.eigh.last_sweeps)
for F, U, w in zip(mats, vecs, vals):
    Fd, Ud = F.double(), U.double()
    Fd = (Fd + Fd.t()) / 2
    res = float(torch.linalg.norm(Fd @ Ud - Ud * w.double()[None, :]) / torch.linalg.norm(Fd))
    orth = float(torch.linalg.norm(Ud.t() @ Ud - torch.eye(U.shape[0], device=dev, dtype=torch.float64)) / U.shape[0] ** 0.5)
    print(tuple(F.shape), "residual %.2e orth %.2e sorted %s" % (res, orth, bool((w[1:] >= w[:-1]).all())))
