#!/bin/bash
# Runs on the GPU box: SQ counters of one script, aggregated per kernel (where the waves' cycles go).
#   usage: tools/pmc_sq.sh <tag> <python script> [args...]      output: gpurun_out/<tag>/sq_summary.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SCRIPT=$1; shift
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS \
    --output-format csv -d "$OUT/pmc_sq" -o run -- python3 "$ROOT/$SCRIPT" "$@" > "$OUT/pmc_sq.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/sq_summary.txt" 2>&1
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void dq::", "").replace("dq::", "").split("<")[0].strip()
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS"]
print(f"{'kernel':30s} {'disp':>5s} " + " ".join(f"{n[3:]:>18s}" for n in names) + "   wait% issue-stall% active% ldsconf/ldsactive")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{k:30s} {len(cnt[k]):5d} " + " ".join(f"{v.get(n, 0):18.0f}" for n in names) +
          f"   {100*v.get('SQ_WAIT_ANY',0)/wc:5.1f} {100*v.get('SQ_WAIT_INST_ANY',0)/wc:5.1f} {100*v.get('SQ_ACTIVE_INST_ANY',0)/wc:5.1f} "
          f"{v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',0),1):5.2f}")
PY
cat "$OUT/sq_summary.txt"
