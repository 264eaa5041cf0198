#!/bin/bash
# Round profiles: run on the GPU box via   gpurun -- 'bash tools/collect_profiles.sh r02'
# 1) rocprofv3 --kernel-trace --stats of bench.py        -> gpurun_out/<round>_bench_kernel_stats.csv
# 2) PMC passes of bench.py (separate passes: TCC slots; the program itself after `--`)
#      FETCH_SIZE | WRITE_SIZE | SQ instruction / MFMA counters | SQ wait counters  -> gpurun_out/<round>_kernels_pmc.json
# 3) the same passes for the eigensolver (the 108 ResNet-50 factors, tools/eigh_r50.py) -> gpurun_out/<round>_eigh_pmc.json
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-other-configs"
# warm MIOpen's per-user find database first: on a fresh box the first run benchmarks every conv solver
$B --steps 1 --warmup 1 > $OUT/warm.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 5 --warmup 2 > $OUT/bench_trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 2 --warmup 1 > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --steps 2 --warmup 1 > $OUT/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- $B --steps 2 --warmup 1 > $OUT/sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- $B --steps 2 --warmup 1 > $OUT/sq2.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
grep '^{"metric"' $OUT/bench_trace.log | tail -1 > gpurun_out/${R}_bench_under_profiler.json
python3 tools/parse_pmc.py $OUT > gpurun_out/${R}_kernels_pmc.json
python3 tools/parse_pmc.py $OUT --factor-build > gpurun_out/${R}_syrk_pmc.json
# eigensolver
E="python3 tools/eigh_r50.py"
EO=$OUT/eigh; mkdir -p $EO
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $EO/trace -- $E > $EO/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $EO/fetch -- $E > $EO/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $EO/write -- $E > $EO/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $EO/sq -- $E > $EO/sq.log 2>&1
cp $(find $EO/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_eigh_kernel_stats.csv
python3 tools/parse_pmc.py $EO jacobi_two_sided32_kernel jacobi_cols_pair32_kernel eigh_norms32_kernel gemm_f64_macro_kernel jacobi_colsv_pair_kernel jacobi_rows_kernel jacobi_cols_kernel > gpurun_out/${R}_eigh_pmc.json
tail -3 $EO/trace.log
grep "factors in\|worst" $EO/trace.log > gpurun_out/${R}_eigh_run.txt
rm -rf $OUT            # raw traces and counter dumps (hundreds of MB): only the summaries above travel back
head -12 gpurun_out/${R}_bench_kernel_stats.csv | cut -c1-150
python3 - <<PY
import json
for f in ("gpurun_out/${R}_kernels_pmc.json", "gpurun_out/${R}_eigh_pmc.json"):
    for k in json.load(open(f)):
        print(k["kernel"], {x: (round(v, 4) if isinstance(v, float) else v) for x, v in k.items() if x in ("mfma_pipe_utilisation", "clock_GHz", "valu_per_mfma", "hbm_bytes_per_launch", "hbm_GBps_fetch_pass", "avg_kernel_ns_sq_pass", "sq_wait_any_share", "sq_wait_inst_any_share")})
PY
