#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h
timeout 900 python -m pytest tests/test_syrk_gpu.py tests/test_kfac_api_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/h/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/h/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/h/bench.json 2> gpurun_out/h/bench.err
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/h/pytest.txt | head -20; cat gpurun_out/h/bench.json
