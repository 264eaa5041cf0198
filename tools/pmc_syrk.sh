cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in 0 3; do
CURV_SYRK_ABLATE=$a rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d gpurun_out/clk_$a -- python tools/bench_syrk.py --model resnet50 --batch 32 --only 3x3s1:2304 --iters 3 > gpurun_out/clk_$a.log 2>&1
tail -1 gpurun_out/clk_$a.log
done
