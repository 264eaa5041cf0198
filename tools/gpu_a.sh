#!/bin/bash
# first GPU pass of a round: tests, fp64 MFMA peak, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/a
./tools/micro/mfma_f64_peak > gpurun_out/a/mfma_peak.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/a/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/a/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/a/bench.json 2> gpurun_out/a/bench.err
tail -5 gpurun_out/a/mfma_peak.txt; tail -40 gpurun_out/a/pytest.txt; cat gpurun_out/a/bench.json
