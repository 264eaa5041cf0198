#!/bin/bash
# kernel trace of the LeNet-5 step (config 2): per-kernel durations and gaps of one steady-state step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lenet
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lenet/trace -- python3 tools/lenet_step.py > gpurun_out/lenet/run.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/lenet/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one steady-state step of the eager loop: from the 30th noise draw back to the previous one
idx = [i for i, r in enumerate(rows) if "randn" in r["Kernel_Name"]]
seg = rows[idx[29] + 1:idx[30] + 1]
seg = seg[-6:] and rows[idx[29]:idx[30]]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
total = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    total += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:64]}  wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}")
    prev_end = max(prev_end, e)
print(f"kernel time of the step (sum of durations; the two MFMA kernels of update() overlap): {total / 1e3:.1f} us")
PY
tail -3 gpurun_out/lenet/run.txt
