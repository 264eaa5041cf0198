#!/bin/bash
# One PMC pass over invert() of the ResNet-50 factors (kernels serialise under counter collection): per kernel the
# launches, the ISOLATED total duration, wave cycles and MFMA-pipe busy share.  Output: gpurun_out/pmc_model.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pim
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pim -- python tools/trace_invert.py 1 > gpurun_out/pim.log 2>&1
python - <<'PY' > gpurun_out/pmc_model.txt
import csv, glob
d = "gpurun_out/pim"
fs = glob.glob(d + "/*/*counter_collection.csv")
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc, tt, cnt, seen = {}, {}, {}, set()
for r in csv.DictReader(open(fs[0])):
    if "curv::" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    acc.setdefault(name, {})
    acc[name][r["Counter_Name"]] = acc[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if (name, r["Dispatch_Id"]) not in seen:
        seen.add((name, r["Dispatch_Id"]))
        tt[name] = tt.get(name, 0) + dur[r["Dispatch_Id"]]
        cnt[name] = cnt.get(name, 0) + 1
calls = 3.0   # trace_invert.py 1: two warm-up calls + one timed
print(f"{'kernel':34s} {'launches':>8s} {'us/call':>9s} {'wave-Mcyc':>10s} {'mfma_busy':>9s} {'Gflop(mfma)':>11s}")
tot = 0.0
for name in sorted(acc, key=lambda n: -tt[n]):
    a = acc[name]
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(1024.0 * gui / 8.0, 1.0)
    tot += tt[name] / 1e3 / calls
    print(f"{name:34s} {cnt[name] / calls:8.0f} {tt[name] / 1e3 / calls:9.1f} {a.get('SQ_WAVE_CYCLES', 0) / 1e6 / calls:10.1f} {busy:9.3f} {a.get('SQ_INSTS_MFMA', 0) / calls / 1e6:11.2f}")
print("sum of isolated kernel time per call (us):", tot)
PY
rm -rf gpurun_out/pim
cat gpurun_out/pmc_model.txt
