#!/bin/bash
# per-queue timeline of one invert() of the ResNet-50 factor sizes (rocprofv3 kernel trace of tools/trace_invert.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trinv
python3 tools/trace_invert.py 5
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trinv/t -- python3 tools/trace_invert.py 1 > gpurun_out/trinv/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trinv/t/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "curv::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last invert call: from the last inv_prepare_kernel on
starts = [i for i, r in enumerate(rows) if "inv_prepare_kernel" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in rows)
print(f"span {(end - t0) / 1e3:.0f} us, {len(rows)} kernels")
qs = {}
for r in rows:
    q = r.get("Queue_Id", "?")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    d = qs.setdefault(q, {"first": s, "last": e, "busy": 0.0, "n": 0, "names": {}})
    d["first"] = min(d["first"], s); d["last"] = max(d["last"], e); d["busy"] += e - s; d["n"] += 1
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    d["names"][n] = d["names"].get(n, 0) + 1
for q, d in sorted(qs.items(), key=lambda kv: kv[1]["first"]):
    print(f"queue {q}: {d['first']:7.0f} .. {d['last']:7.0f} us, busy {d['busy']:7.0f} us in {d['n']} kernels: {d['names']}")
# chain kernels of the large group (the queue with chol_diag and the longest span): gaps between consecutive kernels
PY
