#!/bin/bash
# fp64 MFMA roof: wall time per MFMA + the shader clock during each launch (GRBM_GUI_ACTIVE / duration)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f64roof
./tools/micro/mfma_f64_roof > gpurun_out/f64roof/plain.txt 2>&1
cat gpurun_out/f64roof/plain.txt
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/f64roof/pmc -- ./tools/micro/mfma_f64_roof > gpurun_out/f64roof/pmc.txt 2>&1
python3 - <<'PY'
import csv, glob
ct = glob.glob("gpurun_out/f64roof/pmc/**/*counter_collection.csv", recursive=True)
kt = glob.glob("gpurun_out/f64roof/pmc/**/*kernel_trace.csv", recursive=True)
print(ct, kt)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r.get("Dispatch_Id") or r.get("Correlation_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:40], r.get("Grid_Size"), r.get("Workgroup_Size"))
if ct:
    for r in csv.DictReader(open(ct[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        did = r.get("Dispatch_Id")
        d = dur.get(did)
        v = float(r["Counter_Value"])
        if d: print(did, d[1], "grid", d[2], "wg", d[3], "dur_us %.1f" % (d[0] / 1e3), "GUI_ACTIVE %.0f" % v, "clock GHz (per XCD-summed/8?) %.3f  /8: %.3f" % (v / d[0], v / d[0] / 8))
PY
