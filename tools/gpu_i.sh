#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/i
timeout 900 python -m pytest tests/test_invert_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/i/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/i/pytest.txt
timeout 300 python tools/bench_invert_sizes.py > gpurun_out/i/sizes.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/i/bench.json 2> gpurun_out/i/bench.err
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/i/pytest.txt | head -20; cat gpurun_out/i/sizes.txt; cut -c1-400 gpurun_out/i/bench.json; grep -o '"phases_ms.*' gpurun_out/i/bench.json
