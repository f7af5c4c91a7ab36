#!/usr/bin/env python3
"""Prints the launches of the LAST sort in a rocprofv3 --kernel-trace csv: start offset, duration and the idle gap before
each launch (so a short sort's time can be split into kernel time and launch gaps).
   usage: tools/trace_one_sort.py <dir with *kernel_trace.csv> [first-kernel-substring]"""
import csv, glob, sys

def main():
    path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    first = sys.argv[2] if len(sys.argv) > 2 else "text_hist_kernel"
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    rows = rows[starts[-1]:]
    t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0; busy = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void dq::", "")[:70]
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {name}  grid {r.get('Grid_Size_X', '?')}")
        busy += e - s; prev_end = max(prev_end, e)
    print(f"total {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, {len(rows)} launches")

if __name__ == "__main__":
    main()
