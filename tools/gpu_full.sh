#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/full/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/full/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/full/pytest.txt | head -20; tail -2 gpurun_out/full/smoke.txt; cat gpurun_out/full/bench.json
