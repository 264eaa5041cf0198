#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f
timeout 200 ./tools/micro/flat_syrk_proto 2048 > gpurun_out/f/flat_proto.txt 2>&1
cat gpurun_out/f/flat_proto.txt
