"""Eigensolver on the 108 ResNet-50 factors (N = 32): time, sweeps, worst residual / orthogonality."""
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
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vecs, vals = ops.eigh(mats, with_values=True, tol=tol)
    torch.cuda.synchronize(); print(f"tol {tol}: {len(mats)} factors in {time.perf_counter() - t0:.3f} s, sweeps {ops.eigh.last_sweeps}")
worst = [0.0, 0.0]
for F, U, w in zip(mats, vecs, vals):
    Fd, Ud = F.double(), U.double()
    Fd = (Fd + Fd.t()) / 2
    res = float(torch.linalg.norm(Fd @ Ud - Ud * w.double()[None, :]) / torch.linalg.norm(Fd))
    orth = float(torch.linalg.norm(Ud.t() @ Ud - torch.eye(U.shape[0], device=dev, dtype=torch.float64)) / U.shape[0] ** 0.5)
    worst = [max(worst[0], res), max(worst[1], orth)]
print("worst residual %.2e, worst orthogonality %.2e" % tuple(worst))
