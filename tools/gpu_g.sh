#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g
timeout 1500 python -m pytest tests/test_syrk_gpu.py tests/test_kfac_api_gpu.py tests/test_round2_gpu.py tests/test_fullsize_properties_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/g/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/g/pytest.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/g/bench.json 2> gpurun_out/g/bench.err
timeout 600 python tools/bench_syrk.py --classes > gpurun_out/g/classes.txt 2>&1
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/g/pytest.txt | head -20; cat gpurun_out/g/bench.json; head -16 gpurun_out/g/classes.txt
