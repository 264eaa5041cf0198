#!/bin/bash
# Memory-side fetches of syrk_flat_kernel per class of factor (one ResNet-50 convolution per run, N = 32) against the launch
# plan's operand bytes:   gpurun -- 'bash tools/traffic_by_class.sh'   -> gpurun_out/r06_traffic_by_class.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
O=gpurun_out/r06_traffic_by_class.txt
: > $O
python3 tools/traffic_by_class.py corr64 > /dev/null 2>&1     # (first import of torch on the box)
for C in ${CLASSES:-corr64 corr128 corr256 corr512 unf1152 unf2304 unf4608 g256 a256 a1024 a2048}; do
  python3 tools/traffic_by_class.py $C --model >> $O
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r6/tbc_$C -- python3 tools/traffic_by_class.py $C > gpurun_out/r6/tbc_$C.log 2>&1
  python3 - $C >> $O <<PY
import csv, glob, collections, sys
fs = glob.glob("gpurun_out/r6/tbc_%s/*/*counter_collection.csv" % sys.argv[1])
if not fs:
    print("    no counters"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    if any(k in n for k in ("syrk", "corr_", "prep")):
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(acc.items()):
    f = c["FETCH_SIZE"]
    print("    %-22s launches %2d  fetched %8.1f MB per launch (FETCH_SIZE x 2)" % (n, len(f), 2 * 1024 * sum(f) / len(f) / 1e6))
PY
  rm -rf gpurun_out/r6/tbc_$C
done
cat $O
