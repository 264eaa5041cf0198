"""Compact dump (start_us, end_us, queue, workgroups, kernel) of the LAST complete step of a bench.py kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --no-cpu-baseline --no-other-configs --steps 3 --warmup 2
    python tools/step_trace_dump.py DIR > step.csv"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("curv::", "") for r in rows]
# a step ends with the sampler's gemv_rows launch; the last complete step lies between the last two of them
ends = [i for i, n in enumerate(names) if "gemv_rows" in n]
a, b = ends[-2] + 1, ends[-1]
t0 = int(rows[a]["Start_Timestamp"])
for i in range(a, b + 1):
    r = rows[i]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:.1f},{(int(r['End_Timestamp']) - t0) / 1e3:.1f},{r['Queue_Id']},"
          f"{int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)},{names[i]}")
