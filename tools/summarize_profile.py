#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a small text summary:
per-kernel launch count / average duration from the rocprofv3 kernel trace, and per-kernel
FETCH_SIZE / WRITE_SIZE (PMC, separate passes) per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.split("(")[0]
    for p in ("void dq::", "dq::"):
        name = name.replace(p, "")
    return name.split("<")[0].strip()


def main():
    out = sys.argv[1]
    print(f"# profile summary for {os.path.basename(out)}")
    stats = find(os.path.join(out, "stats"), "*kernel_stats.csv")
    if stats:
        print("\n## rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
        print(f"{'kernel':34s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
        for r in csv.DictReader(open(stats)):
            name = short(r["Name"])
            print(f"{name:34s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} "
                  f"{float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} "
                  f"{float(r['Percentage']):6.2f}")
    trace = find(os.path.join(out, "stats"), "*kernel_trace.csv")
    if trace:
        per = defaultdict(list)
        for r in csv.DictReader(open(trace)):
            per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                 r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""),
                                                 r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
        print("\n## kernel_trace.csv (per kernel: VGPRs, LDS bytes, grid, workgroup of the last launch)")
        for k, v in per.items():
            print(f"{k:34s} launches={len(v):5d} vgpr={v[-1][1]:>4s} lds={v[-1][2]:>6s} grid={v[-1][3]:>9s} wg={v[-1][4]:>5s}")
    traffic = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = find(os.path.join(out, f"pmc_{counter}"), "*counter_collection.csv")
        if not f:
            continue
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print(f"\n## PMC {counter} (rocprofv3 --pmc {counter}; unit KiB as reported; per launch)")
        for k, (cnt, tot) in agg.items():
            print(f"{k:34s} launches={cnt:5d} {counter}_per_launch_KiB={tot/cnt:14.1f}  (= {tot/cnt*1024/1e6:10.2f} MB)")
            traffic.setdefault(k, {})[counter + "_KiB_per_launch"] = tot / cnt
            traffic[k]["launches_" + counter] = cnt
    if traffic:
        import json
        for k, v in traffic.items():
            if "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
                # gfx950: FETCH_SIZE reports 1/2 of the bytes of coalesced streaming reads
                # (MI355X_MICROARCH.md, HBM section; calibrated here on kernels with known reads)
                v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KiB_per_launch"] + v["WRITE_SIZE_KiB_per_launch"]) * 1024
        if len(sys.argv) > 2:
            traffic["workload"] = sys.argv[2]            # bench.py reports traffic only for the workload it was measured on
        # ... and only for the library the counters were taken from: the digest of its sources at profiling time
        try:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from deltaq_amd import build as dq_build
            traffic["library_source_digest"] = dq_build._source_digest()
        except Exception as e:                            # noqa: BLE001
            traffic["library_source_digest"] = None
            print("no source digest:", e)
        json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
