#!/usr/bin/env python3
"""Text-like inputs of the reference's benchmark sizes through the HOST interface (what bench.py's text_like_sizes
reports), under forced settings: median of 30 calls per variant, variants interleaved three times.
usage: t_abi_small.py "" "DQ_TAIL_MAX=0" ..."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deltaq_amd import HipSuffixSort
from tools import datagen

s = HipSuffixSort(0)
variants = sys.argv[1:] or [""]
for size in (16384, 65536, 262144, 1048576, 4194304):
    T = datagen.gen_enwik_like(size, 0xD17A0, 65536)
    sa = np.ones(size, np.int32)
    s.Sort(T, sa)
    res = {v: [] for v in variants}
    for rep in range(3):
        for v in variants:
            sets = dict(kv.split("=") for kv in v.split(",") if kv)
            for k, val in sets.items(): os.environ[k] = val
            s.Sort(T, sa)
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); s.Sort(T, sa); ts.append(time.perf_counter() - t0)
            res[v].append(sorted(ts)[len(ts) // 2] * 1e6)
            for k in sets: del os.environ[k]
    print(size, {v or "defaults": [round(x, 1) for x in r] for v, r in res.items()}, flush=True)
