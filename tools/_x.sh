cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in asm4 asmfill; do
  rm -rf gpurun_out/q; mkdir -p gpurun_out/q
  CURV_ALT_LIB=tools/micro/libcurv_$m.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q/trace -- python3 tools/ab_update.py > gpurun_out/q/log.txt 2>&1
  f=$(find gpurun_out/q/trace -name "*kernel_stats.csv" | head -1)
  echo "== $m"
  python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("curv::", "")
    if any(k in n for k in ("corr_",)):
        print(f"  {n:32s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
rm -rf gpurun_out/q
