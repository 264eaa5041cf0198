# far outer_update_kernel time per ablation variant (tools/ablate_outer.py build must have run in the container)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1 2 3; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/abl$v -- python tools/ablate_outer.py run $v > gpurun_out/abl$v.log 2>&1
done
python - <<'PY'
import csv, glob
for v in range(4):
    fs = glob.glob(f"gpurun_out/abl{v}/*/*kernel_trace.csv")
    if not fs:
        print(v, "no trace"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "curv::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) * 2 // 3:]
    span = (max(int(r["End_Timestamp"]) for r in rows) - int(rows[0]["Start_Timestamp"])) / 1e3
    far = [r for r in rows if "outer_update" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 256 * 1200]
    near = [r for r in rows if "outer_update" in r["Kernel_Name"] and int(r["Grid_Size_X"]) <= 256 * 1200]
    d = lambda rs: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
    big = max(far, key=lambda r: int(r["Grid_Size_X"]))
    print(f"variant {v}: call span {span:.0f} us, far {len(far)} launches {d(far):.0f} us, other outer {len(near)} launches {d(near):.0f} us; largest far grid {int(big['Grid_Size_X']) // 256} wgs {(int(big['End_Timestamp']) - int(big['Start_Timestamp'])) / 1e3:.0f} us")
PY
