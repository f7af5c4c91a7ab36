#!/bin/bash
# Copies the summaries of tools/profile_gpu.sh runs (gpurun_out/<tag>/) into profiles/<round>/ (tracked).
#   usage: tools/collect_profiles.sh r02 r02b_uniform256 r02b_enwik256 ...
set -eu
ROUND=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/profiles/$ROUND"
for TAG in "$@"; do
  SRC=$ROOT/gpurun_out/$TAG
  cp "$SRC/summary.txt" "$ROOT/profiles/$ROUND/${TAG}_summary.txt"
  cp "$SRC/traffic.json" "$ROOT/profiles/$ROUND/${TAG}_traffic.json"
  STATS=$(find "$SRC/stats" -name '*kernel_stats.csv' | head -1)
  [ -n "$STATS" ] && cp "$STATS" "$ROOT/profiles/$ROUND/${TAG}_kernel_stats.csv"
  # (the line printed under rocprof, when the profiled command was bench.py: no empty file otherwise)
  if grep -q '^{' "$SRC/stats.log" 2>/dev/null; then
    grep '^{' "$SRC/stats.log" | tail -1 > "$ROOT/profiles/$ROUND/${TAG}_bench_line_under_rocprof.json"
  fi
done
