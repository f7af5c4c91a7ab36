#!/usr/bin/env python3
"""Bucketed round 0 (dq_bucket_sort.h) on a GPU box: parity on uniform / forced / fallback inputs, then timing
against the plain digit passes.   python tests/manual/t_bucket.py [quick]"""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from deltaq_amd import HipSuffixSort, _abi
from tools import datagen

L = _abi.load(); s = HipSuffixSort(0)

def launches(name):
    return _abi.profile_snapshot()[name]["launches"]

def check(T, tag, expect_bucket=None):
    L.dq_profile_reset(); L.dq_profile_enable(1)
    sa = s.Sort(T)
    L.dq_profile_enable(0)
    nb = launches("bucket_sort_kernel")
    ok = np.array_equal(sa, oracle.divsufsort(T)) if T.size <= (80 << 20) else oracle.sufcheck_mt(T, sa) == 0
    print(f"{tag:44s} n={T.size:10d} bucket launches={nb} rank launches={launches('radix_rank_kernel')} {'OK' if ok else 'WRONG'}", flush=True)
    assert ok, tag
    if expect_bucket is not None:
        assert (nb > 0) == expect_bucket, (tag, nb)

rnd = datagen.gen_uniform
check(rnd(13_000_000, 1), "uniform 13 MB (auto)", True)
check(rnd((64 << 20) + 777, 2), "uniform 64 MiB+777 (auto)", True)
os.environ["DQ_BUCKET"] = "1"
for n in (70_000, 300_001, 1 << 20, 3_000_000):
    check(rnd(n, n), f"uniform forced n={n}")
check(rnd(2_000_000, 5) & 0x3F, "6-bit alphabet forced (fallback or not)")
T = rnd(6_000_000, 7); T[1_000_000:1_060_000] = 0x41
check(T, "run of 60000 x 0x41 forced (fallback)")
T = rnd(6_000_000, 8); T[100:5100] = T[3_000_000:3_005_000]; T[-9:] = 0
check(T, "repeat of 5000 + zero tail forced")
check(np.zeros(200_000, np.uint8), "all zeros forced (fallback)")
del os.environ["DQ_BUCKET"]
T = rnd(8_000_000, 9); T[::7] = 0
check(T, "1/7 zeros 8 MB (auto: model decides)")
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)

def timeit(T, reps=10):
    dT = torch.from_numpy(T).cuda(); out = torch.empty(T.size, dtype=torch.int32, device="cuda")
    s.Sort(dT, out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): s.Sort(dT, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    L.dq_profile_reset(); L.dq_profile_enable(1)
    for _ in range(3): s.Sort(dT, out)
    torch.cuda.synchronize(); L.dq_profile_enable(0)
    prof = {k: round(v["ms"] / 3, 3) for k, v in _abi.profile_snapshot().items() if v["launches"]}
    return dt * 1e3, prof

for mib, seed in ((64, 0x5EED0002), (256, 0x5EED0003), (16, 0x5EED0500)):
    T = rnd(mib << 20, seed)
    os.environ["DQ_NO_BUCKET"] = "1"
    ms, prof = timeit(T)
    print(f"{mib} MiB plain passes      : {ms:7.3f} ms  {prof}", flush=True)
    del os.environ["DQ_NO_BUCKET"]
    ms, prof = timeit(T)
    print(f"{mib} MiB bucketed round 0 : {ms:7.3f} ms  {prof}", flush=True)
