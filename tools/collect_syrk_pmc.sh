#!/bin/bash
# Only the factor-build traffic file of a round (bench.py's `roofline.traffic`): the FETCH_SIZE / WRITE_SIZE / SQ passes of
# tools/collect_profiles.sh and the --factor-build summary.   gpurun -- 'bash tools/collect_syrk_pmc.sh r06'
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_syrk_$R
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-other-configs"
$B --steps 1 --warmup 1 > $OUT/warm.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B --steps 2 --warmup 1 > $OUT/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $B --steps 2 --warmup 1 > $OUT/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- $B --steps 2 --warmup 1 > $OUT/sq.log 2>&1
python3 tools/parse_pmc.py $OUT --factor-build > gpurun_out/${R}_syrk_pmc.json
rm -rf $OUT
python3 -c "
import json; d = json.load(open('gpurun_out/${R}_syrk_pmc.json')); print(d['source_sha16'], '%.2f GB per update' % (d['hbm_bytes_per_launch'] / 1e9))"
