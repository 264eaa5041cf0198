# LDS / wait counters for the grouped SYRK kernel (diagnostics)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_THREAD_CYCLES[A-Z_0-9]*" | sort -u > gpurun_out/avail_lds.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcL -- python tools/bench_syrk.py --model resnet50 --batch 32 --iters 2 > gpurun_out/pmcL.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcM -- python tools/bench_syrk.py --model resnet50 --batch 32 --iters 2 > gpurun_out/pmcM.log 2>&1
python - <<'PY'
import csv, glob
for d in ("gpurun_out/pmcL", "gpurun_out/pmcM"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); continue
    acc, n = {}, {}
    for r in csv.DictReader(open(fs[0])):
        if "syrk_patch_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
    for k in acc:
        print(d, k, acc[k] / n[k])
PY
tail -2 gpurun_out/pmcL.log gpurun_out/pmcM.log
