# L2 hit/miss + fabric reads of syrk_patch_kernel for one factor class:  bash tools/pmc_class.sh 1x1s1:1024
C=${1:-1x1s1:1024}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pc1 -- python tools/bench_syrk.py --only $C --iters 2 > gpurun_out/pc1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pc2 -- python tools/bench_syrk.py --only $C --iters 2 > gpurun_out/pc2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/pc3 -- python tools/bench_syrk.py --only $C --iters 2 > gpurun_out/pc3.log 2>&1
python - <<'PY'
import csv, glob
for d in ("gpurun_out/pc1", "gpurun_out/pc2", "gpurun_out/pc3"):
    fs = glob.glob(d + "/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); continue
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
    acc, n, t = {}, {}, {}
    for r in csv.DictReader(open(fs[0])):
        if "syrk_patch_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
        t[r["Dispatch_Id"]] = dur[r["Dispatch_Id"]]
    print(d, {k: f"{acc[k] / n[k]:.4g}" for k in acc}, "avg kernel us", sum(t.values()) / max(len(t), 1) / 1e3)
PY
tail -1 gpurun_out/pc1.log
