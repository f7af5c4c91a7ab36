#!/bin/bash
# PMC passes over a python script (GPU box).  usage: tools/pmc_py.sh <tag> <kernel-substring> script.py [args]
TAG=$1; KSUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o run -- python3 "$ROOT/$1" "${@:2}" > $OUT/p$i.log 2>&1
done
cd $ROOT
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, ksub = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][:80]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print("==", k)
    for c, (n, tot) in sorted(d.items()):
        print(f"   {c:32s} per-launch {tot/n:16.1f}   (launches {n})")
PY
