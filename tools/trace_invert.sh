cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/trace_invert.py 5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trinv -- python tools/trace_invert.py 1 > gpurun_out/trinv.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trinv/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
# last of the three invert calls (2 warm-up + 1 timed)
rows = [r for r in rows if "curv::" in r["Kernel_Name"] and "inv_" in r["Kernel_Name"] or "chol_" in r["Kernel_Name"] or "_update_kernel" in r["Kernel_Name"] or "panel_product" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) * 2 // 3:]
t0 = int(rows[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in rows)
print("span us", (end - t0) / 1e3, "kernels", len(rows))
busy = {}
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    busy[n] = busy.get(n, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print({k: round(v) for k, v in busy.items()})
# overlap: time covered by outer_update far kernels (largest grids) vs chain kernels
for r in rows[:0]:
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f}  q{r.get("Queue_Id", "?")} grid {r.get("Grid_Size", r.get("Grid_Size_X", "?"))}  {n}')
PY
