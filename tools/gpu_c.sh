#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c
timeout 1500 python -m pytest tests/test_syrk_gpu.py tests/test_kfac_api_gpu.py tests/test_efb_inf_gpu.py tests/test_sharding_gpu.py tests/test_round2_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/c/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/c/pytest.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/c/bench.json 2> gpurun_out/c/bench.err
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/c/pytest.txt | head -20; cat gpurun_out/c/bench.json
