#!/bin/bash
# FETCH_SIZE calibration on the factor-build kernels' access pattern (tools/micro/fetch_calib.hip):
#   gpurun -- 'bash tools/fetch_calib.sh'   -> gpurun_out/r06_fetch_calib.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
O=gpurun_out/r06_fetch_calib.txt
./tools/micro/fetch_calib > $O 2>&1
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  T=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/r6/calib_$T -- ./tools/micro/fetch_calib > gpurun_out/r6/calib_$T.log 2>&1
  python3 - "$T" >> $O <<PY
import csv, glob, collections, sys
fs = glob.glob("gpurun_out/r6/calib_%s/*/*counter_collection.csv" % sys.argv[1])
if not fs:
    print("no counter file for", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"].split("(")[0]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(acc.items()):
    print(n, {k: "%.4g (mean of %d)" % (sum(v) / len(v), len(v)) for k, v in c.items()})
PY
  rm -rf gpurun_out/r6/calib_$T
done
cat $O
