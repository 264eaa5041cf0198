#!/bin/bash
# L2-miss (FETCH_SIZE) traffic of the factor-build kernels in a short bench.py run
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fq; mkdir -p gpurun_out/fq
python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fq/fetch -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/fq/fetch.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/fq/fetch/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "syrk" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in acc.items():
    print(f"{k}: FETCH_SIZE {v / n * 1024 / 1e9 * 1.0:.3f} GB per launch raw-KiB*1024 ({n} launches; gfx950 correction as in tools/parse_pmc.py not applied)")
PY
