#!/bin/bash
# kernel statistics of INF.invert at ResNet-50 size (per call, from four calls of tools/prof_inf_invert.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/inf
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/inf/trace -- python3 tools/prof_inf_invert.py > gpurun_out/inf/run.log 2>&1
grep "inf.invert\|ab sizes" gpurun_out/inf/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/inf/trace/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "jacobi" not in r["Name"] and "eigh" not in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"]) / 4e6:9.2f} ms per invert  {int(r["Calls"]) // 4:>6} calls  {r["Name"][:80]}')
PY
