#!/bin/bash
# Round profiles: run on the GPU box via   gpurun -- 'bash tools/collect_profiles.sh r01'
# 1) rocprofv3 --kernel-trace --stats of bench.py        -> gpurun_out/<round>_bench_kernel_stats.csv
# 2) PMC passes (separate: TCC slots) FETCH_SIZE / WRITE_SIZE / MFMA counters -> gpurun_out/<round>_syrk_pmc.json
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
# warm MIOpen's per-user find database first: on a fresh box the first run benchmarks every conv solver
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/warm.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
grep '^{"metric"' $OUT/bench_trace.log | tail -1 > gpurun_out/${R}_bench_under_profiler.json
python tools/parse_pmc.py $OUT > gpurun_out/${R}_syrk_pmc.json
cat gpurun_out/${R}_syrk_pmc.json
head -8 gpurun_out/${R}_bench_kernel_stats.csv | cut -c1-140
