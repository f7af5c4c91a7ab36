#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of tests/manual/t_case.py <case> [variants...], kernels listed by
# their full names (template arguments tell the geometries of one kernel apart).   usage: tools/prof_case.sh <tag> <case> [variants]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$ROOT/tests/manual/t_case.py" "$@" > "$OUT/stats.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/kernels.txt"
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "stats", "**", "*kernel_stats.csv"), recursive=True)
for r in csv.DictReader(open(f[0])):
    name = r["Name"].replace("void dq::", "").replace("dq::", "")
    name = name.split("(")[0][:90]
    print(f"{name:92s} calls={int(r['Calls']):5d} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.2f}")
PY
cat "$OUT/kernels.txt"
