#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/d
for t in 1024 4096; do echo "== target items $t"; timeout 120 ./tools/micro/flat_syrk_proto $t; done > gpurun_out/d/flat_proto.txt 2>&1
timeout 600 python tools/bench_syrk.py --classes > gpurun_out/d/classes.txt 2>&1
bash tools/pmc_valu_per_mfma.sh 3x3s1:2304 3x3s1:576 3x3s1:4608 1x1s1:1024 G:1024 1x1s1:256 3x3s1:1152 > gpurun_out/d/valu.txt 2>&1
cat gpurun_out/d/flat_proto.txt gpurun_out/d/classes.txt gpurun_out/d/valu.txt
