#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b
for t in 512 1024 2048; do echo "== target items $t"; timeout 120 ./tools/micro/flat_syrk_proto $t; done > gpurun_out/b/flat_proto.txt 2>&1
timeout 1500 python -m pytest tests/test_round2_gpu.py tests/test_sharding_gpu.py tests/test_efb_inf_gpu.py "tests/test_kfac_api_gpu.py::test_sharded_efb_inf_diagonal_cover_the_unsharded_result" "tests/test_fullsize_properties_gpu.py::test_config5_resnet50_inf_chain" -m gpu -q --tb=short -s -p no:cacheprovider > gpurun_out/b/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/b/pytest.txt
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/b/bench.json 2> gpurun_out/b/bench.err
cat gpurun_out/b/flat_proto.txt; grep -E "passed|failed|FAILED|rc=|P_c|INF own|config 5|Error" gpurun_out/b/pytest.txt | head -40; cat gpurun_out/b/bench.json
