#!/usr/bin/env python3
"""Developer check on a GPU box: HIP path vs the oracle on a ladder of inputs, with timings.
(Test infrastructure; the judged parity tests are tests/test_gpu_parity.py.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from deltaq_amd import HipSuffixSort, _abi  # noqa: E402

have_dss = hasattr(oracle.lib(), "dq_oracle_divsufsort_i32")


def check(name, T, s, dtype=np.int32):
    T = np.ascontiguousarray(T, dtype=np.uint8)
    t0 = time.time()
    SA = s.Sort(T, index_dtype=dtype)
    dt = time.time() - t0
    info = _abi.last_sort_info()
    ok = True
    if T.size <= 40000:
        ref = oracle.naive_sa(T).astype(dtype)
        ok = np.array_equal(SA, ref)
    elif have_dss and T.size <= (64 << 20):
        ref = oracle.divsufsort(T, dtype=dtype)
        ok = np.array_equal(SA, ref)
    else:
        ok = oracle.sufcheck(T, SA) == 0 and oracle.verify_sampled(T, SA, 200000, 5) == -1
    print(f"{'OK ' if ok else 'BAD'} {name:28s} n={T.size:>10d} {dt*1e3:9.2f} ms  {info}", flush=True)
    if not ok:
        bad = np.nonzero(SA != ref)[0] if T.size <= 40000 else []
        print("   first mismatches:", bad[:10], SA[bad[:5]] if len(bad) else "")
    return ok


def main():
    big = "--big" in sys.argv
    s = HipSuffixSort(0)
    allok = True
    allok &= check("shruggy", np.frombuffer("¯\\_(ツ)_/¯".encode(), dtype=np.uint8), s)
    for n in (3, 4, 7, 8, 9, 63, 64, 65, 1000, 4095, 4096, 4097, 8193, 20000):
        allok &= check(f"rnd{n}", oracle.net_random_bytes(n), s)
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "tests/golden/assets/*"))):
        allok &= check(os.path.basename(f)[:20], np.fromfile(f, dtype=np.uint8), s)
    for n in (3, 8, 9, 65, 1025, 4097, 10000):
        allok &= check(f"zeros{n}", np.zeros(n), s)
        allok &= check(f"ab{n}", np.tile([97, 98], n)[:n], s)
        allok &= check(f"rnd{n}+zeros", np.concatenate([oracle.net_random_bytes(n), np.zeros(9)]), s)
    allok &= check("enwik20k", oracle.gen_enwik_like(20000, 0xD17A0, 4096), s)
    allok &= check("enwik20k/i64", oracle.gen_enwik_like(20000, 0xD17A0, 4096), s, np.int64)
    allok &= check("uniform 1M", oracle.gen_uniform(1 << 20, 2), s)
    allok &= check("enwik 1M", oracle.gen_enwik_like(1 << 20, 0xD17A0, 65536), s)
    allok &= check("uniform 16M", oracle.gen_uniform(16 << 20, 0x5EED0500), s)
    allok &= check("uniform 16M/i64", oracle.gen_uniform(16 << 20, 0x5EED0500), s, np.int64)
    allok &= check("enwik 16M", oracle.gen_enwik_like(16 << 20, 0xD17A0, 262144), s)
    if big:
        allok &= check("uniform 64M", oracle.gen_uniform(64 << 20, 0x5EED0002), s)
        allok &= check("uniform 256M", oracle.gen_uniform(256 << 20, 0x5EED0003), s)
        allok &= check("enwik 256M", oracle.gen_enwik_like(256 << 20, 0xD17A0, 262144), s)
    # timing with profiling on a resident buffer
    import torch
    L = _abi.load()
    for n in ((64 << 20,) if not big else (64 << 20, 256 << 20)):
        T = torch.from_numpy(oracle.gen_uniform(n, 0x5EED0002)).cuda()
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        s.Sort(T, out)
        torch.cuda.synchronize()
        L.dq_profile_enable(1); L.dq_profile_reset()
        t0 = time.time()
        for _ in range(3):
            s.Sort(T, out)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        L.dq_profile_enable(0)
        print(f"device-resident {n>>20} MiB: {dt*1e3:.2f} ms/sort = {n/1e6/dt:.1f} MB/s")
        for k, v in _abi.profile_snapshot().items():
            if v["launches"]:
                print(f"   {k:28s} launches={v['launches']:4d} total={v['ms']:9.3f} ms avg={v['ms']/v['launches']*1e3:9.1f} us "
                      f"alg={v['alg_bytes']/max(v['ms'],1e-9)/1e6:8.1f} GB/s")
    print("ALL OK" if allok else "FAILURES")
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
