# PMC passes over the invert of three 4608^2 factors (kernels serialise under counter collection: isolated
# per-kernel durations + L2 / MFMA / LDS counters of outer_update_kernel)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cat > /tmp/big3.py <<'PY'
import sys, torch
sys.path.insert(0, '.')
from curvature_amd import ops
dev = torch.device('cuda:0')
Fs = []
for i in range(3):
    torch.manual_seed(i)
    X = torch.randn(4608, 4096, device=dev)
    Fs.append((X @ X.t() / 4096).contiguous())
for _ in range(2):
    ops.chol_inv_lower(Fs, [1.0] * 3, [1000.0] * 3, check=False)
torch.cuda.synchronize()
PY
timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pi1 -- python /tmp/big3.py > gpurun_out/pi1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pi2 -- python /tmp/big3.py > gpurun_out/pi2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pi3 -- python /tmp/big3.py > gpurun_out/pi3.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pi4 -- python /tmp/big3.py > gpurun_out/pi4.log 2>&1
python - <<'PY'
import csv, glob
for d in ("gpurun_out/pi1", "gpurun_out/pi2", "gpurun_out/pi3", "gpurun_out/pi4"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); continue
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
    acc, tt, cnt = {}, {}, {}
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"].split("(")[0].replace("curv::", "")
        if "curv::" not in r["Kernel_Name"]:
            continue
        if name == "outer_update_kernel":
            name += "_far" if int(r.get("Grid_Size_X") or r["Grid_Size"]) > 256 * 1200 else "_small"
        acc.setdefault(name, {})
        acc[name][r["Counter_Name"]] = acc[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if (name, r["Dispatch_Id"]) not in seen:
            seen.add((name, r["Dispatch_Id"]))
            tt[name] = tt.get(name, 0) + dur[r["Dispatch_Id"]]
            cnt[name] = cnt.get(name, 0) + 1
    for name in acc:
        print(d, name, "launches", cnt[name], "total_us", tt[name] / 1e3, {k: f"{v:.4g}" for k, v in acc[name].items()})
PY
tail -2 gpurun_out/pi1.log
