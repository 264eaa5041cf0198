# per-kernel time of the eigensolver on one 2304^2 factor
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/eig1.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from curvature_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
n, r = 2304, 1000
X = torch.randn(n, r, device=dev) * torch.logspace(0, -3, r, device=dev)
F = X @ X.t() / r
F = ((F + F.t()) / 2).contiguous()
ops.eigh([F])
torch.cuda.synchronize()
print("sweeps", ops.eigh.last_sweeps)
PY
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tre -- python /tmp/eig1.py > gpurun_out/tre.log 2>&1
grep sweeps gpurun_out/tre.log
grep "curv::" $(find gpurun_out/tre -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-120
