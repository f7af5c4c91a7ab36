#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST sort in a rocprofv3 kernel_trace.csv (start offset, duration, gap)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.split("(")[0].replace("void dq::", "").replace("dq::", "")
    return n.split("<")[0][:34]
# find last text_hist_kernel as the beginning of the last sort
idx = max(i for i, r in enumerate(rows) if "text_hist_kernel" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
tot = 0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}  gap {(s-prev_end)/1e3:7.1f}  {short(r['Kernel_Name'])}")
    prev_end = e; tot += e - s
print(f"span {(prev_end-t0)/1e3:.1f} us, kernel time {tot/1e3:.1f} us")
