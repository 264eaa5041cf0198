#!/bin/bash
# MFMA-pipe share and clock of the factor-build kernels inside bench.py's step (one SQ counter pass; kernels serialise under
# counter collection):   gpurun -- 'bash tools/pmc_flat_quick.sh [tag]'
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
B="python3 bench.py --no-cpu-baseline --no-other-configs"
$B --steps 1 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/r6/pmc_$TAG -- $B --steps 2 --warmup 1 > gpurun_out/r6/pmc_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/r6/pmc_$TAG/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    if not any(k in n for k in ("syrk", "corr_", "prep")): continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], n)
    if key not in seen:
        seen.add(key); cnt[n] += 1
        acc[n]["dur"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for n, c in acc.items():
    k = cnt[n]
    gui = c["GRBM_GUI_ACTIVE"] / 8 / k            # cycles per launch (sum over 8 XCDs)
    dur = c["dur"] / k
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / k / (256 * 4) / max(gui, 1)
    print("%-22s launches %3d  %8.1f us  clock %.2f GHz  mfma busy %.3f  mfma %9.0f  valu/mfma %.2f salu/mfma %.2f" % (
        n, k, dur / 1e3, gui / max(dur, 1), busy, c["SQ_INSTS_MFMA"] / k, c["SQ_INSTS_VALU"] / max(c["SQ_INSTS_MFMA"], 1), c["SQ_INSTS_SALU"] / max(c["SQ_INSTS_MFMA"], 1)))
PY
rm -rf gpurun_out/r6/pmc_$TAG
