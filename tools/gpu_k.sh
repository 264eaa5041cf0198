#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k
timeout 600 python tools/fullchain_resnet18.py resnet50 32 > gpurun_out/k/fullchain50.txt 2>&1
timeout 600 python tools/fullchain_resnet18.py resnet18 32 > gpurun_out/k/fullchain18.txt 2>&1
timeout 300 python tools/bench_configs.py > gpurun_out/k/configs.txt 2>&1
timeout 600 python tools/emulate_sharding.py > gpurun_out/k/emulate.txt 2>&1
timeout 300 python tools/bench_invert_sizes.py > gpurun_out/k/sizes.txt 2>&1
cat gpurun_out/k/fullchain50.txt gpurun_out/k/fullchain18.txt gpurun_out/k/configs.txt gpurun_out/k/emulate.txt gpurun_out/k/sizes.txt
