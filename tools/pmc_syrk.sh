# PMC passes for the grouped SYRK kernel on the ResNet-50 workload (separate passes: TCC slots)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d gpurun_out/pmcA -- python tools/bench_syrk.py --model resnet50 --batch 32 --iters 2 > gpurun_out/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- python tools/bench_syrk.py --model resnet50 --batch 32 --iters 2 > gpurun_out/pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- python tools/bench_syrk.py --model resnet50 --batch 32 --iters 2 > gpurun_out/pmcW.log 2>&1
tail -1 gpurun_out/pmcA.log
