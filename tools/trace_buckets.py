#!/usr/bin/env python
"""Per time bucket of the LAST call in a rocprofv3 kernel trace: busy time per kernel name (overlaps add up) and queue."""
import csv
import glob
import sys
from collections import defaultdict

root, bucket_us = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
f = glob.glob(root + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "curv::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: everything that starts after the previous call's last finalize pass ended (`per_call` finalize launches
# per call: 2 for a whole model - large and small group -, 1 for a chain-bound call)
per_call = int(sys.argv[3]) if len(sys.argv) > 3 else 2
fin = sorted(int(r["End_Timestamp"]) for r in rows if "inv_finalize" in r["Kernel_Name"])
calls = len(fin) // per_call
cut = fin[-per_call - 1] if len(fin) > per_call else 0
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
t0 = int(rows[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in rows)
print(f"span {(end - t0) / 1e3:.1f} us, {len(rows)} kernels, {calls} calls")
names = sorted({r["Kernel_Name"].split("(")[0].replace("curv::", "").replace("_kernel", "") for r in rows})
nb = int((end - t0) / 1e3 / bucket_us) + 1
busy = [defaultdict(float) for _ in range(nb)]
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "").replace("_kernel", "")
    st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    b = int(st / bucket_us)
    while b * bucket_us < en:
        lo, hi = max(st, b * bucket_us), min(en, (b + 1) * bucket_us)
        busy[b][n] += hi - lo
        b += 1
print("bucket_start_us " + " ".join(f"{n[:12]:>12s}" for n in names))
for b in range(nb):
    print(f"{b * bucket_us:10.0f}      " + " ".join(f"{busy[b][n] / bucket_us:12.2f}" for n in names))
# which queues carried what (streams that share a hardware queue serialise)
qs = defaultdict(lambda: defaultdict(lambda: [0, 1e18, 0.0]))
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("curv::", "").replace("_kernel", "")
    e = qs[r.get("Queue_Id", "?")][n]
    e[0] += 1
    e[1] = min(e[1], (int(r["Start_Timestamp"]) - t0) / 1e3)
    e[2] = max(e[2], (int(r["End_Timestamp"]) - t0) / 1e3)
for q, d in sorted(qs.items()):
    print(f"queue {q}: " + ", ".join(f"{n} x{c} [{a:.0f}..{b:.0f}]" for n, (c, a, b) in sorted(d.items())))
