#!/bin/bash
# kernel timeline of INF.sample_and_replace at ResNet-50 size (the last call of tools/prof_inf_sample.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6; rm -rf gpurun_out/r6/infs
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6/infs -- python3 tools/prof_inf_sample.py > gpurun_out/r6/infs.log 2>&1
grep "sample_and_replace" gpurun_out/r6/infs.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r6/infs/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last call: behind the last but one randn launch
idx = [i for i, r in enumerate(rows) if "randn" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r.get("Grid_Size_X", 0) or 0) // max(int(r.get("Workgroup_Size_X", 256) or 256), 1)
    print(f"{(st - t0) / 1e3:9.1f} dur {(en - st) / 1e3:8.1f} wgs {wg:6d}  {n}")
PY
rm -rf gpurun_out/r6/infs
