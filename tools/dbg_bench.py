import sys, torch
sys.path.insert(0, '.')
from curvature_amd import models, ops
from curvature_amd.curvatures import KFAC
import oracle.curvature_oracle as o
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = models.resnet50().to(dev).train()
kfac = KFAC(model)
layers = kfac._layers()
x = torch.randn(32, 3, 224, 224, device=dev)
logits = model(x)
labels = torch.distributions.Categorical(logits=logits.detach()).sample()
loss = torch.nn.functional.cross_entropy(logits, labels)
model.zero_grad(); loss.backward()
for it in range(2):
    kfac.update(batch_size=32)
    torch.cuda.synchronize()
    for li in (25, 29, 34, 2):
        A, G = kfac.state[layers[li]]
        xin, g = kfac.record[layers[li]]
        ref = None
        if it == 0:
            geom = o.layer_geometry(layers[li])
            import torch.nn.functional as F
            if layers[li].kernel_size != (1, 1) or True:
                cols = F.unfold(xin.detach().double(), layers[li].kernel_size, padding=layers[li].padding, stride=layers[li].stride)
                X = cols.permute(1, 0, 2).reshape(cols.shape[1], -1)
                ref = X @ X.t() / X.shape[1]
        ev = torch.linalg.eigvalsh(A.double())
        msg = f"it{it} layer{li} A dim {A.shape[0]} finite {bool(torch.isfinite(A).all())} sym {bool(torch.equal(A, A.t()))} absmax {float(A.abs().max()):.3e} eig min {float(ev[0]):.3e} max {float(ev[-1]):.3e} xin max {float(xin.abs().max()):.2e} contiguous {xin.is_contiguous()} g max {float(g.abs().max()):.2e}"
        if ref is not None:
            msg += f" relerr {float((A.double()-ref).norm()/ref.norm()):.2e}"
        print(msg)
try:
    kfac.invert(1.0, 1000.0)
    print("invert ok")
except RuntimeError as e:
    print("invert failed:", str(e)[:200])
