#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes of ONE
# python script (a config that is not bench.py's workload, e.g. the 2 GiB int64 sort).  Output: gpurun_out/<tag>/
# in the layout tools/summarize_profile.py and tools/collect_profiles.sh expect.
#   usage: tools/profile_script.sh <tag> <workload-name> script.py [args...]
set -u
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- \
    python3 "$ROOT/$1" "${@:2}" > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o run -- \
      python3 "$ROOT/$1" "${@:2}" > "$OUT/pmc_$C.log" 2>&1
done
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" "$WL" > "$OUT/summary.txt" 2>&1
tail -5 "$OUT/stats.log" >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
