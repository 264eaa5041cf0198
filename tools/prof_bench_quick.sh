#!/bin/bash
# kernel stats of a short bench.py run (rocprofv3 --kernel-trace --stats), curv:: kernels only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
python3 bench.py --no-cpu-baseline --no-other-configs --steps 1 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/quick/trace -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 > gpurun_out/quick/trace.log 2>&1
f=$(find gpurun_out/quick/trace -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/quick/kernel_stats.csv
grep -E "syrk|corr|upload_table" $f | cut -c1-170
grep '^{"metric"' gpurun_out/quick/trace.log | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'], d['roofline']['kernel_ms'])"
