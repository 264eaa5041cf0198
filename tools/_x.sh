for m in none after before; do echo -n "$m: "; python bench.py --stream-probe $m --batch 32 2>/dev/null | tail -1; done
for o in a01m m01a xxam01 xxxam01; do echo -n "$o none: "; CURV_STREAM_ORDER=$o python bench.py --stream-probe none --batch 32 2>/dev/null | tail -1; done
python tools/bench_invert_sizes.py 2>&1 | grep -v amdgpu | tail -12
timeout 900 python -m pytest tests/test_invert_gpu.py tests/test_estimator_chain_gpu.py tests/test_graph_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
