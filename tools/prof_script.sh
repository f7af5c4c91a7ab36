#!/bin/bash
# usage: tools/prof_script.sh <tag> script.py [args]  -> kernel stats by name (GPU box)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 "$ROOT/$1" "${@:2}" > $OUT/stats.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
print(f"{'kernel':40s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0].replace("void dq::", "").replace("dq::", "")[:40]
    print(f"{name:40s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")
PY
