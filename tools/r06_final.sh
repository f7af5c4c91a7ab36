#!/bin/bash
# Runs on the GPU box (one gpurun call): the profiles and the bench line of the final round-6 build.
#   usage: tools/r06_final.sh <prefix>     output under gpurun_out/<prefix>_*
set -u
P=${1:-r06z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
bash tools/profile_gpu.sh ${P}_uniform256 > /dev/null 2>&1
DQ_PROFILE_WORKLOAD=enwik256 bash tools/profile_gpu.sh ${P}_enwik256 --workload enwik > /dev/null 2>&1
bash tools/profile_script.sh ${P}_libtorch128 libtorch128 tests/manual/t_case.py libtorch128 > /dev/null 2>&1
bash tools/pmc_sq.sh ${P}_sq_enwik256 bench.py --workload enwik --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-build --no-profile > /dev/null 2>&1
mkdir -p gpurun_out/${P}
(time python bench.py --config3-cpu) > gpurun_out/${P}/bench_line_config3_cpu.json 2> gpurun_out/${P}/bench_stderr.log
tail -c 600 gpurun_out/${P}/bench_stderr.log
head -c 1500 gpurun_out/${P}/bench_line_config3_cpu.json
for t in uniform256 enwik256 libtorch128; do echo "== $t"; head -30 gpurun_out/${P}_$t/summary.txt; done
cat gpurun_out/${P}_sq_enwik256/sq_summary.txt
