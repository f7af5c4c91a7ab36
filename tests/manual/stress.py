#!/usr/bin/env python3
"""Long randomised differential run on a GPU box (not collected by pytest):
    python tests/manual/stress.py [seconds] [seed]
Structured random inputs of 1 B .. 24 MB (tests/test_gpu_parity.py::structured_text), random forced-path
environments, int32 / int64, host and device entry points; every SA is bit-compared with the oracle."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle
from deltaq_amd import HipSuffixSort
from structured_inputs import structured_text
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ENVS = [{}, {}, {}, {"DQ_SMALL_N": "0"}, {"DQ_NO_FUSED_TIES": "1"}, {"DQ_NO_SMALL": "1"}, {"DQ_NO_BINNED_ISA": "1"},
        {"DQ_NO_CHAIN": "1"}, {"DQ_PACKED": "1", "DQ_KEY_BYTES": "2"}, {"DQ_PACKED": "1", "DQ_KEY_BYTES": "3"},
        {"DQ_PACKED": "0", "DQ_KEY_BYTES": "4"}, {"DQ_SPARSE": "1"}, {"DQ_SPARSE": "0"},
        {"DQ_CODED": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8"}, {"DQ_CODED": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_SMALL_N": "0"},
        {"DQ_PAIR_CHAINS": "1"}, {"DQ_PAIR_CHAINS": "2", "DQ_SMALL_N": "0"}, {"DQ_PAIR_CHAINS": "2", "DQ_PAIR_MAXG": "4"},
        {"DQ_PAIR_CHAINS": "1", "DQ_PAIR_MAXG": "3", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2"},
        {"DQ_PAIR_CHAINS": "2", "DQ_PAIR_MAXG": "2", "DQ_NO_BINNED_ISA": "1"}, {"DQ_BUCKET": "1"}, {"DQ_NO_FIRST_SMALL": "1"},
        {"DQ_BINNED_ISA": "1"}, {"DQ_BINNED_ISA": "1", "DQ_NO_FIRST_SMALL": "1"}, {"DQ_BINNED_ISA": "1", "DQ_RUNS": "1", "DQ_SMALL_N": "0"},
        {"DQ_RUNS": "1"}, {"DQ_RUNS": "1", "DQ_SMALL_N": "0"}, {"DQ_RUNS": "1", "DQ_NO_SMALL": "1"}, {"DQ_RUNS": "1", "DQ_SPARSE": "0", "DQ_PAIR_CHAINS": "2"},
        {"DQ_RUNS": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_MID_GROUPS": "256"}, {"DQ_RUNS": "0"},
        {"DQ_MID_GROUPS": "0"}, {"DQ_MID_GROUPS": "0", "DQ_SMALL_N": "0"}, {"DQ_MID_GROUPS": "256"}, {"DQ_MID_GROUPS": "512", "DQ_NO_CHAIN": "1"},
        {"DQ_MID_GROUPS": "1024", "DQ_SMALL_N": "0"}, {"DQ_MID_GROUPS": "256", "DQ_PAIR_CHAINS": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2"},
        {"DQ_MID_GROUPS": "1024", "DQ_PAIR_CHAINS": "0", "DQ_PACKED": "1", "DQ_KEY_BYTES": "2", "DQ_SMALL_N": "0"},
        {"DQ_LATE_RUNS_MIN": "1"}, {"DQ_LATE_RUNS_MIN": "1", "DQ_SMALL_N": "0"}, {"DQ_LATE_RUNS_MIN": "16", "DQ_MID_GROUPS": "256"},
        {"DQ_LATE_RUNS_MIN": "1", "DQ_RUN_PERIOD": "1"}, {"DQ_LATE_RUNS_MIN": "1", "DQ_RUN_PERIOD": "3", "DQ_SMALL_N": "0"},
        {"DQ_LATE_RUNS_MIN": "4", "DQ_RUN_PERIOD": "8", "DQ_PAIR_CHAINS": "0"},
        {"DQ_LATE_RUNS_MIN": "1", "DQ_UPD_BIN_MIN": "1", "DQ_PAIR_CHAINS": "2"}, {"DQ_NO_LATE_RUNS": "1"},
        {"DQ_UPD_BIN": "2", "DQ_UPD_BIN_MIN": "1"}, {"DQ_UPD_BIN": "1", "DQ_UPD_BIN_MIN": "1", "DQ_SMALL_N": "0"},
        {"DQ_UPD_BIN": "2", "DQ_UPD_BIN_MIN": "1", "DQ_RUNS": "1", "DQ_SMALL_N": "0"}, {"DQ_NO_UPD_WORDS": "1"},
        {"DQ_SPARSE": "1", "DQ_BINNED_ISA": "1"}, {"DQ_SPARSE": "1", "DQ_BINNED_ISA": "1", "DQ_SMALL_N": "0", "DQ_LATE_RUNS_MIN": "1"},
        # round 5: lists of more than n/2 tied suffixes through the LDS class (third list buffer), the XCD-aware first pass
        {"DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"}, {"DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_SMALL_N": "0"},
        {"DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_BINNED_ISA": "1", "DQ_UPD_BIN_MIN": "1", "DQ_SMALL_N": "0"},
        {"DQ_PACKED": "0", "DQ_KEY_BYTES": "3", "DQ_SPARSE": "0", "DQ_RUNS": "1", "DQ_NO_UPD_WORDS": "1"},
        {"DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_MID_GROUPS": "0"}, {"DQ_NO_WIDE_SMALL": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
        {"DQ_NO_L_SHIFT": "1"}, {"DQ_NO_L_SHIFT": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_MID_GROUPS": "256"},
        {"DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_MID_GROUPS": "256", "DQ_PAIR_CHAINS": "0", "DQ_SMALL_N": "0"},
        {"DQ_UPD_WINDOW": "1"}, {"DQ_UPD_WINDOW": "1", "DQ_SMALL_N": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
        {"DQ_UPD_WINDOW": "1", "DQ_RUNS": "1", "DQ_LATE_RUNS_MIN": "1"}, {"DQ_UPD_WINDOW": "0"},
        {"DQ_TAIL_MAX": "0"}, {"DQ_TAIL_MAX": "0", "DQ_SMALL_N": "0"}, {"DQ_TAIL_MAX": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
        {"DQ_TAIL_MAX": "0", "DQ_RUNS": "1"}, {"DQ_TAIL_MAX": "0", "DQ_MID_GROUPS": "0"}, {"DQ_TAIL_MAX": "0", "DQ_PAIR_CHAINS": "2"},
        {"DQ_TAIL_MAX": "77", "DQ_SMALL_N": "0"}, {"DQ_TAIL_MAX": "4096", "DQ_RUNS": "1", "DQ_SMALL_N": "0"},
        {"DQ_TAIL_MAX": "2000", "DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_SMALL_N": "0"},
        {"DQ_CHAIN_STEPS": "1"}, {"DQ_CHAIN_STEPS": "1", "DQ_TAIL_MAX": "0", "DQ_SMALL_N": "0"},
        {"DQ_CHAIN_STEPS": "3", "DQ_TAIL_MAX": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
        {"DQ_CHAIN_STEPS": "3", "DQ_TAIL_MAX": "0", "DQ_MID_GROUPS": "256", "DQ_SMALL_N": "0"},
        {"DQ_XCD_GROUP": "0"}, {"DQ_XCD_GROUP": "3", "DQ_SMALL_N": "0"}, {"DQ_XCD_GROUP": "64", "DQ_BUCKET": "1"},
        # round 6: round 0 as a sample sort (dq_split_round0.h), forced from 4 MiB on: raw / coded keys, past the heavy-key check
        {"DQ_SPLIT": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8"}, {"DQ_SPLIT": "2", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_CODED": "1"},
        {"DQ_SPLIT": "2", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_CODED": "0", "DQ_BINNED_ISA": "1"},
        {"DQ_SPLIT": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_CODED": "1", "DQ_RUNS": "1"}]
KEYS = sorted({k for e in ENVS for k in e})
s = HipSuffixSort(0)
t_end = time.time() + budget
count = 0; total = 0
while time.time() < t_end:
    u = rng.random()
    n = int(rng.integers(1, 3000)) if u < 0.25 else int(rng.integers(3000, 200_000)) if u < 0.7 else \
        int(rng.integers(200_000, 3_000_000)) if u < 0.95 else int(rng.integers(3_000_000, 24_000_000))
    env = ENVS[int(rng.integers(0, len(ENVS)))]
    if "DQ_SPLIT" in env:                               # (the path takes texts of >= 5 MiB)
        n = int(rng.integers(5_300_000, 9_000_000))
    T = structured_text(rng, n)
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
    ref = oracle.divsufsort(T)
    mode = int(rng.integers(0, 4))
    if os.environ.get("STRESS_LOG"):
        with open(os.environ["STRESS_LOG"], "w") as f:
            f.write(repr(dict(count=count, n=n, env=env, mode=mode)) + "\n")
        np.save(os.environ["STRESS_LOG"] + ".npy", T)
    if mode == 0:   got = s.Sort(T)
    elif mode == 1: got = s.Sort(T, index_dtype=np.int64)
    elif mode == 2: got = s.Sort(torch.from_numpy(T).cuda()).cpu().numpy()
    else:           got = s.Sort(torch.from_numpy(T).cuda(), index_dtype=np.int64).cpu().numpy()
    if not np.array_equal(got.astype(np.int64), ref.astype(np.int64)):
        np.save(os.path.join(ROOT, "gpurun_out", f"stress_fail_{seed}_{count}.npy"), T)
        print("MISMATCH", dict(count=count, n=n, env=env, mode=mode), flush=True)
        sys.exit(1)
    count += 1; total += n
print(f"stress OK: {count} inputs, {total/1e6:.1f} MB, seed {seed}", flush=True)
