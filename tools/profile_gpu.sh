#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for
# HBM traffic of one bench.py invocation.  Output: gpurun_out/<tag>/...
#   usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- \
    python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-build "$@" > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o run -- \
      python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-build --no-profile "$@" > "$OUT/pmc_$C.log" 2>&1
done
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" "${DQ_PROFILE_WORKLOAD:-uniform-256MiB}" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
