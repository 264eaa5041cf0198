#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j
timeout 1200 python -m pytest tests/test_invert_gpu.py tests/test_efb_inf_gpu.py tests/test_round2_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/j/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/j/pytest.txt
timeout 300 python tools/bench_invert_sizes.py > gpurun_out/j/sizes.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/j/bench.json 2> gpurun_out/j/bench.err
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/j/pytest.txt | head -20; cat gpurun_out/j/sizes.txt; grep -o '"ms_per_step[^,]*' gpurun_out/j/bench.json; grep -o '"phases_ms.*' gpurun_out/j/bench.json
