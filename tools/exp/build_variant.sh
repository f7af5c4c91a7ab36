#!/bin/bash
# Experiment builds: the library with one -D switch, as tools/exp/libdq_<name>.so (git-ignored; travels with gpurun).
#   usage: tools/exp/build_variant.sh <name> -DDQ_EXPERIMENT_...   then   DQ_SUFSORT_LIB=tools/exp/libdq_<name>.so python ...
# The timing switches themselves (DQ_EXPERIMENT_MG_NOGATHER / _MG_NOWALK / _SKIP_LOOKBACK, DQ_EXP_STOP_ROUNDS) are not in
# the shipped kernels: `git apply -p1 tools/exp/timing_experiments.patch` puts them into a scratch copy of the tree first
# (round-4 numbers: DESIGN.md section 5; tests/manual/t_exp_round1.py drives them).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/deltaq_amd/csrc
O=$ROOT/tools/exp/obj_$NAME
mkdir -p "$O"
for s in dq_sorter_i32 dq_sorter_i64 dq_diff dq_abi; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -pthread "$@" -c $C/$s.hip -o $O/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -pthread $O/*.o -o $ROOT/tools/exp/libdq_$NAME.so
rm -rf "$O"
echo built tools/exp/libdq_$NAME.so
