cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from curvature_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
X = torch.randn(4608, 4096, device=dev)
F = (X @ X.t() / 4096).contiguous()
for _ in range(3):
    ops.chol_inv_lower([F], [1.0], [1000.0], check=False)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr1 -- python /tmp/one.py > gpurun_out/tr1.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tr1/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "curv::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * 2 // 3:]
t0 = int(rows[0]["Start_Timestamp"])
print("span us", (max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e3, "kernels", len(rows))
prev = t0
# panels 8..9 (mid sweep)
start = None
cnt = 0
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    if n in ("chol_diag_kernel", "chol_square_kernel"):
        cnt += 1
    if cnt in (9, 10):
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f'{(st - t0) / 1e3:9.1f} dur {(en - st) / 1e3:7.1f}  q{r.get("Queue_Id", "?")} wgs {int(r.get("Grid_Size_X", 0)) // max(int(r.get("Workgroup_Size_X", 256)), 1):6d}  {n}')
PY
