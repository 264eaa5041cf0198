#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box by tools/gpu_full.sh and tools/collect_profiles.sh) -> profiles/<round>_*
R=${1:-r06}
cd "$(dirname "$0")/.."
cp gpurun_out/${R}_bench_kernel_stats.csv profiles/${R}_bench_kernel_stats_full.csv
(head -1 gpurun_out/${R}_bench_kernel_stats.csv; grep '^"curv::' gpurun_out/${R}_bench_kernel_stats.csv) > profiles/${R}_bench_kernel_stats_curv.csv
(head -1 gpurun_out/${R}_eigh_kernel_stats.csv; grep '^"curv::' gpurun_out/${R}_eigh_kernel_stats.csv) > profiles/${R}_eigh_kernel_stats_curv.csv
cp gpurun_out/${R}_bench_under_profiler.json gpurun_out/${R}_kernels_pmc.json gpurun_out/${R}_syrk_pmc.json gpurun_out/${R}_eigh_pmc.json gpurun_out/${R}_eigh_run.txt profiles/
tail -1 gpurun_out/full/bench.json > profiles/${R}_bench.json
