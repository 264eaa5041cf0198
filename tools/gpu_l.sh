#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/l
timeout 1500 python -m pytest tests/test_kfac_api_gpu.py tests/test_efb_inf_gpu.py tests/test_round2_gpu.py tests/test_sharding_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/l/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/l/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/l/bench.json 2> gpurun_out/l/bench.err
grep -E "passed|failed|FAILED|rc=|Error|config 5" gpurun_out/l/pytest.txt | head -20; grep -o '"ms_per_step[^,]*' gpurun_out/l/bench.json; grep -o '"phases_ms.*' gpurun_out/l/bench.json
