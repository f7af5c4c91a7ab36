#!/usr/bin/env python3
"""The sample-sort round 0 (dq_split_round0.h) forced on small inputs, one case after the other, with the phase trace
(DQ_TRACE=2 synchronises after every phase): where does a failing case die?   usage: t_split_small.py [case substring]"""
import os, sys
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")
os.environ.setdefault("DQ_SPLIT", "1")
os.environ.update({"DQ_PACKED": "0", "DQ_KEY_BYTES": "8"})
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from deltaq_amd import HipSuffixSort
from tools import datagen
rnd = datagen.gen_uniform
rep = rnd(1_200_000, 5)
cases = {
    "uniform 6 MB": rnd(6_000_000, 0x5EED0002),
    "text 9 MiB": datagen.gen_enwik_like(9 << 20, 21, 65536),
    "text 5 MiB + 5": datagen.gen_enwik_like((5 << 20) + 5, 22, 16384),
    "16 symbols": rnd(5_500_000, 8) & 15,
    "2 symbols": rnd(5_300_000, 9) & 1,
    "zeros": np.zeros((5 << 20) + 1, np.uint8),
    "repeats + zero tail": np.concatenate([rep, rnd(1_500_000, 6), rep[:400_000], rep, rnd(1_300_000, 7), np.zeros(13, np.uint8)]),
}
s = HipSuffixSort(0)
want = sys.argv[1] if len(sys.argv) > 1 else ""
for name, T in cases.items():
    if want not in name:
        continue
    T = np.ascontiguousarray(T, dtype=np.uint8)
    print("==", name, T.size, flush=True)
    import time
    os.environ["DQ_TRACE"] = "2"
    t0 = time.perf_counter()
    sa = s.Sort(T)
    print(f"   first sort {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    del os.environ["DQ_TRACE"]
    t0 = time.perf_counter()
    sa = s.Sort(T)
    print(f"   second sort {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    ok = np.array_equal(sa, oracle.divsufsort(T))
    print("   bit-exact:", ok, flush=True)
    if not ok:
        bad = np.nonzero(sa != oracle.divsufsort(T))[0]
        print("   first differences at", bad[:8], "of", bad.size, flush=True)
