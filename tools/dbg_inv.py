import sys, torch
sys.path.insert(0, '.')
from curvature_amd import ops
import oracle.curvature_oracle as o
dev = torch.device('cuda:0')
def spd(n, seed):
    torch.manual_seed(seed)
    X = torch.randn(n, n + 10)
    F = (X @ X.t() / X.shape[1]).float()
    return (F + F.t()) / 2
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
for sizes in ([1024], [1088], [2304], [1024, 2304, 64, 2048], [64] * 70 + [1100]):
    Fs = [spd(n, i) for i, n in enumerate(sizes)]
    try:
        outs = ops.chol_inv_lower([F.to(dev) for F in Fs], [0.5] * len(Fs), [2.0] * len(Fs), check=False)
        torch.cuda.synchronize()
        errs = []
        for F, L in zip(Fs, outs):
            reg = (torch.tensor(2.0 ** 0.5) * F + torch.diag(F.new_full((F.shape[0],), 0.5 ** 0.5))).double()
            errs.append(rel(L, o.chol_of_inverse((reg + reg.t()) / 2)))
        print(sizes[:6], len(sizes), "max err %.2e" % max(errs), ["%.1e" % e for e in errs[:6]], "last %.1e" % errs[-1])
    except Exception as ex:
        print(sizes[:6], "EXC", ex)
