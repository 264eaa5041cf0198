#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t
timeout 600 python tools/emulate_sharding.py > gpurun_out/t/emulate.txt 2>&1; tail -14 gpurun_out/t/emulate.txt
timeout 300 python tools/bench_configs.py > gpurun_out/t/configs.txt 2>&1; tail -6 gpurun_out/t/configs.txt
timeout 600 python tools/fullchain_resnet18.py resnet50 32 > gpurun_out/t/fullchain50.txt 2>&1; grep -v "^ " gpurun_out/t/fullchain50.txt | tail -24
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t/inf -- python3 tools/prof_inf_invert.py > gpurun_out/t/inf.log 2>&1
grep "inf.invert\|ab sizes" gpurun_out/t/inf.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/t/inf/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms  {r["Calls"]:>6} calls  {r["Name"][:90]}')
PY
