import sys, time, torch
sys.path.insert(0, '.')
from curvature_amd import ops
dev = torch.device('cuda:0')
def psd(n, r, seed):
    torch.manual_seed(seed)
    X = torch.randn(n, r, device=dev) * torch.logspace(0, -3, r, device=dev)
    F = X @ X.t() / r
    return ((F + F.t()) / 2).contiguous()
for n, r in [(576, 300), (1152, 2000), (2304, 1000), (4608, 1568)]:
    F = psd(n, r, n)
    for tol in (1e-6, 1e-8, 1e-10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        (U,), (w,) = ops.eigh([F], with_values=True, tol=tol, max_sweeps=20)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        Ud, Fd = U.double(), F.double()
        res = float(torch.linalg.norm(Fd @ Ud - Ud * w.double()) / torch.linalg.norm(Fd))
        orth = float(torch.linalg.norm(Ud.t() @ Ud - torch.eye(n, device=dev, dtype=torch.float64)) / n ** 0.5)
        print(f"n={n} rank={r} tol={tol:g}: sweeps {ops.eigh.last_sweeps} time {dt*1e3:.0f} ms residual {res:.1e} orth {orth:.1e}", flush=True)
