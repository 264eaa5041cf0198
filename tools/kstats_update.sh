#!/bin/bash
# per-kernel times of update() alone on a model, for a tree (default: this one):  bash tools/kstats_update.sh densenet121 [tree]
M=${1:-resnet50}; T=${2:-.}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
D=gpurun_out/r6/ks_$$
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $T/tools/update_only.py $M 30 > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$D/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("syrk", "corr_", "prep", "upload")):
        print("$M $T %-24s calls %4s avg %9.1f us total %9.1f us/update" % (n.split("(")[0].replace("curv::", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3 / 35))
PY
rm -rf $D
