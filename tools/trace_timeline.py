#!/usr/bin/env python
"""Kernel timeline of the LAST of three identical calls in a rocprofv3 --kernel-trace csv (tools/trace_sizes.sh):
start (us), duration (us), queue, workgroups, kernel - lines [lo, hi) - and per-kernel totals."""
import csv
import glob
import sys
from collections import defaultdict

root, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(root + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "curv::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: everything that starts after the previous call's last finalize pass ended
per_call = int(sys.argv[4]) if len(sys.argv) > 4 else 1
fin = sorted(int(r["End_Timestamp"]) for r in rows if "inv_finalize" in r["Kernel_Name"])
cut = fin[-per_call - 1] if len(fin) > per_call else 0
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
t0 = int(rows[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in rows)
print(f"span {(end - t0) / 1e3:.1f} us, {len(rows)} kernels")
tot = defaultdict(lambda: [0, 0.0])
for i, r in enumerate(rows):
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "")
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot[n][0] += 1
    tot[n][1] += (en - st) / 1e3
    if lo <= i < hi:
        wg = int(r.get("Grid_Size_X", 0) or 0) // max(int(r.get("Workgroup_Size_X", 256) or 256), 1)
        print(f"{i:5d} {(st - t0) / 1e3:9.1f} dur {(en - st) / 1e3:7.1f} q{r.get('Queue_Id', '?'):>3} wgs {wg:6d}  {n}")
print("totals (count, us):")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:36s} {c:5d} {us:10.1f}")
