import sys, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from curvature_amd import ops
dev = torch.device('cuda:0')
def check(N, C, H, k, s, p, bias=False):
    torch.manual_seed(1)
    x = torch.relu(torch.randn(N, C, H, H, device=dev))
    cols = F.unfold(x.double(), k, padding=p, stride=s)
    X = cols.permute(1, 0, 2).reshape(cols.shape[1], -1)
    if bias:
        X = torch.cat([X, torch.ones_like(X[:1])], 0)
    ref = X @ X.t() / X.shape[1]
    n = X.shape[0]
    A = torch.full((n, n), float('nan'), device=dev)
    ops.kfac_accumulate([ops.FactorJob(x, A, (k, k), (s, s), (p, p), bias, 1.0 / X.shape[1], True)])
    torch.cuda.synchronize()
    err = (A.double() - ref).norm() / ref.norm()
    bad = (~torch.isfinite(A)).sum().item()
    # per 128-tile error map
    T = (n + 127) // 128
    worst = []
    for ti in range(T):
        for tj in range(T):
            blk = (A.double() - ref)[ti*128:(ti+1)*128, tj*128:(tj+1)*128]
            e = blk.norm() / (ref[ti*128:(ti+1)*128, tj*128:(tj+1)*128].norm() + 1e-30)
            if e > 1e-4 or not torch.isfinite(e):
                worst.append((ti, tj, float(e)))
    print(f"N={N} C={C} H={H} k={k} s={s}: dim={n} K={X.shape[1]} rel err {float(err):.2e} nonfinite {bad} bad tiles {len(worst)} {worst[:8]}")
check(8, 256, 14, 3, 1, 1)
check(8, 128, 28, 3, 1, 1)
check(32, 256, 14, 3, 1, 1)
check(32, 1024, 14, 1, 1, 0)
check(32, 256, 28, 3, 2, 1)
check(4, 512, 7, 3, 1, 1)
check(32, 512, 7, 3, 1, 1)
check(32, 2048, 1, 1, 1, 0, True)
