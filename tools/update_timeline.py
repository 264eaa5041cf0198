"""One update() of the headline workload as a timeline: every kernel between two consecutive inv_prepare launches' predecessors.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2
    python tools/update_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("curv::", "") for r in rows]
# an update() starts with the first syrk-side kernel after a sampler's gemv / gemm_nt and ends at the first inv_prepare
# the last complete update(): from the first kernel behind a sampler's last launch to the first inv_prepare behind it
samp = [i for i, n in enumerate(names) if "gemm_nt" in n or "gemv_rows" in n]
preps = [i for i, n in enumerate(names) if "inv_prepare" in n]
start = max(i for i in samp if any(p > i for p in preps)) + 1
end = min(p for p in preps if p > start)
t0 = int(rows[start]["Start_Timestamp"])
for i in range(start, end + 1):
    r = rows[i]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(st - t0) / 1e3:9.1f} .. {(en - t0) / 1e3:9.1f}  dur {(en - st) / 1e3:8.1f}  q{r['Queue_Id']:>3} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):>6}  {names[i][-40:]}")
