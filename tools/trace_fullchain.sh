# kernel statistics of the full estimator chain on ResNet-18 (where INF update / invert spend their time)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/fullchain_resnet18.py resnet18 8 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trfc -- python tools/fullchain_resnet18.py resnet18 8 > gpurun_out/trfc.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trfc/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"]) / 1e6:9.2f} ms  {r["Calls"]:>6} calls  {r["Name"][:90]}')
PY
grep "inf\.\|efb\.\|ctor" gpurun_out/trfc.log
