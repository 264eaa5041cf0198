"""GPU-idle gaps inside the LAST occurrence of a window of a kernel trace: the window runs from the first kernel whose name
contains START (searching back from the end for the last such run) to the last kernel whose name contains END.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/prof_inf_invert.py
    python tools/idle_gaps.py DIR clamp_min0 gemm_f64 [min_gap_us]
Prints the window, its busy time (union over all queues) and every gap above the threshold with the kernels on either side."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
start_key, end_key = sys.argv[2], sys.argv[3]
thr = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("curv::", "") for r in rows]
b = max(i for i, n in enumerate(names) if end_key in n)
starts = [i for i, n in enumerate(names) if start_key in n and i < b]
# the last run of START kernels before b: walk back while consecutive START hits are close
a = starts[-1]
for i in reversed(starts[:-1]):
    if int(rows[a]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 2_000_000 and not any(end_key in names[k] for k in range(i, a)):
        a = i
t0 = int(rows[a]["Start_Timestamp"])
iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, names[i]) for i, r in enumerate(rows[a:b + 1], a))
end = max(e for _, e, _ in iv)
busy, cur_e, last = 0, 0, ""
gaps = []
for s_, e_, n in iv:
    if s_ > cur_e:
        if cur_e and s_ - cur_e > thr * 1e3: gaps.append((cur_e, s_, last, n))
        busy += 0
    if e_ > cur_e:
        busy += e_ - max(s_, cur_e)
        cur_e, last = e_, n
print(f"window {end / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(end - busy) / 1e3:.1f} us, {b - a + 1} kernels")
for g0, g1, p, n in gaps:
    print(f"  {g0 / 1e3:10.1f} .. {g1 / 1e3:10.1f}  {(g1 - g0) / 1e3:7.1f} us   after {p[-36:]:36s} before {n[-36:]}")
