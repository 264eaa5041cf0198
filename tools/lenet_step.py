"""LeNet-5 N=100 KFAC step in a loop (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curvature_amd import models
from curvature_amd.curvatures import KFAC
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = models.lenet5().to(dev).eval()
k = KFAC(model)
x = torch.rand(100, 1, 28, 28, device=dev)
logits = model(x)
labels = torch.distributions.Categorical(logits=logits.detach()).sample()
torch.nn.functional.cross_entropy(logits, labels).backward()
def step():
    k.update(100); k.invert(0.5, 1); k.sample_and_replace()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 50 * 1e3)
for name, fn in (("update", lambda: k.update(100)), ("invert", lambda: k.invert(0.5, 1)), ("sample", k.sample_and_replace)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 50 * 1e3)
from curvature_amd.graph import KFACStepGraph
g = KFACStepGraph(k, add=0.5, multiply=1, batch_size=100)
for _ in range(10): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); print("graph replay ms/step", (time.perf_counter() - t0) / 200 * 1e3)
g.check()
