import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "inv_finalize" in r["Kernel_Name"])
# find the last-but-one finalize to show a full step's tail + next update start
fins = [i for i, r in enumerate(rows) if "inv_finalize" in r["Kernel_Name"]]
i0 = fins[-3] if len(fins) >= 3 else fins[0]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
n = 0
for r in rows[i0:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-50:]
    if "copyBuffer" in name and (en - st) < 20000:
        n += 1
        prev_end = en
        continue
    print(f"{(st - t0) / 1e3:9.1f} dur {(en - st) / 1e3:8.1f} gap {(st - prev_end) / 1e3:7.1f}  {name}" + (f"   [{n} small copies before]" if n else ""))
    n = 0
    prev_end = en
    if "syrk_patch" in name:
        break
