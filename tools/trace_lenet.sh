#!/bin/bash
# kernel trace of the LeNet-5 step (config 2): per-kernel durations and gaps of one steady-state step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lenet
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lenet/trace -- python3 tools/lenet_step.py > gpurun_out/lenet/run.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/lenet/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step: find the last upload / first kernel of update: take the last 40 kernels
rows = rows[-45:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size')}")
    prev_end = max(prev_end, e)
PY
tail -3 gpurun_out/lenet/run.txt
