cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_INST[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" | sort -u | tr "\n" " "
echo
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcI -- python tools/bench_syrk.py --model resnet50 --batch 32 --iters 2 > gpurun_out/pmcI.log 2>&1
python - <<'PY'
import csv, glob
for d in ("gpurun_out/pmcI",):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); print(open("gpurun_out/pmcI.log").read()[-1500:]); continue
    acc, n = {}, {}
    for r in csv.DictReader(open(fs[0])):
        if "syrk_patch_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
    for k in acc:
        print(d, k, acc[k] / n[k])
PY
