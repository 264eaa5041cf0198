# VALU / SALU / LDS instructions per MFMA of syrk_patch_kernel for factor classes (instruction accounting,
# LAB_NOTEBOOK.md section 8 item 0):  bash tools/pmc_valu_per_mfma.sh 3x3s1:2304 1x1s1:1024 G:1024
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for C in "$@"; do
  rm -rf gpurun_out/pv
  timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pv -- python tools/bench_syrk.py --only $C --iters 2 > gpurun_out/pv.log 2>&1
  python - "$C" <<'PY'
import csv, glob, sys
fs = glob.glob("gpurun_out/pv/*/*counter_collection.csv")
if not fs:
    print(sys.argv[1], "no counters"); sys.exit(0)
acc, n = {}, {}
for r in csv.DictReader(open(fs[0])):
    if "syrk_patch_kernel" not in r["Kernel_Name"]:
        continue
    acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
a = {k: acc[k] / n[k] for k in acc}
m = a.get("SQ_INSTS_MFMA", 1.0)
print(f'{sys.argv[1]}: per MFMA: VALU {(a.get("SQ_INSTS_VALU", 0) - m) / m:.2f} (MFMA excluded if counted), SALU {a.get("SQ_INSTS_SALU", 0) / m:.2f}, '
      f'LDS {a.get("SQ_INSTS_LDS", 0) / m:.2f}, VMEM reads {a.get("SQ_INSTS_VMEM_RD", 0) / m:.3f}; MFMA busy '
      f'{a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * a.get("GRBM_GUI_ACTIVE", 1) / 8.0) * 100:.1f} % ; raw VALU/MFMA {a.get("SQ_INSTS_VALU", 0) / m:.2f}')
PY
done
