cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/eprof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/eprof/t -- python3 tools/eigh_r50.py > gpurun_out/eprof/log.txt 2>&1
grep "factors in" gpurun_out/eprof/log.txt
grep "curv::" $(find gpurun_out/eprof/t -name "*kernel_stats.csv" | head -1) | sed 's/(.*)"/"/' | cut -d, -f1-4 | head -12
