#!/bin/bash
# quick look at the factor build on the GPU box: parity of the factor-build tests, the bench line's phases, per-kernel times
#   gpurun -- 'bash tools/prof_update_quick.sh [tag] [notest]'
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
if [ "$2" != "notest" ]; then
  timeout 900 python -m pytest tests/test_syrk_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
fi
B="python3 bench.py --no-cpu-baseline --no-other-configs"
for i in 1 2; do
$B --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']
print('step %.2f update %.3f invert %.2f sample %.3f window %.3f frac %.3f' % (d['ms_per_step'], p['update'], p['invert'], p['sample_and_replace'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/trace_$TAG -- $B --steps 5 --warmup 2 > gpurun_out/r6/bench_trace_$TAG.log 2>&1
cp $(find gpurun_out/r6/trace_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/r6/kstats_$TAG.csv
rm -rf gpurun_out/r6/trace_$TAG
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/r6/kstats_$TAG.csv")):
    n = r["Name"]
    if any(k in n for k in ("syrk", "corr_", "patch_prep")):
        print("%-28s calls %4s avg %9.1f us  min %9.1f max %9.1f" % (n.split("(")[0].replace("curv::", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
