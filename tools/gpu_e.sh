#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e
timeout 120 ./tools/micro/flat_syrk_proto 1024 > gpurun_out/e/flat_proto.txt 2>&1
rm -rf gpurun_out/e/p1 gpurun_out/e/p2
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d gpurun_out/e/p1 -- ./tools/micro/flat_syrk_proto 1024 > gpurun_out/e/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/e/p2 -- ./tools/micro/flat_syrk_proto 1024 > gpurun_out/e/p2.log 2>&1
python - <<'PY' > gpurun_out/e/pmc.txt
import csv, glob
for d in ("p1", "p2"):
    fs = glob.glob(f"gpurun_out/e/{d}/*/*counter_collection.csv")
    if not fs:
        print(d, "no counters"); continue
    rows = list(csv.DictReader(open(fs[0])))
    # per dispatch: keep dispatches of flat_syrk_kernel; group by Grid_Size as the class key
    acc = {}
    for r in rows:
        if "flat_syrk" not in r["Kernel_Name"]:
            continue
        key = (r["Grid_Size"],)
        acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for key, c in acc.items():
        print(d, key, {k: sum(v) / len(v) for k, v in c.items()})
PY
cat gpurun_out/e/flat_proto.txt gpurun_out/e/pmc.txt
