"""Latency of the host API (ISuffixSort.Sort(text, suffixes)) at the sizes the reference's own
benchmark uses (SuffixSortingBenchmarks.cs:27-53: 64 B .. 32 KiB, 64 KiB .. 1 MiB)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaq_amd import HipSuffixSort, workload
s = HipSuffixSort(0)
for n in (64, 256, 1024, 4096, 16384, 32768, 65536, 262144, 1 << 20, 4 << 20):
    for name in ("uniform", "period2"):
        if name == "uniform":
            T = workload.gen_uniform(n, 5)
        else:
            T = np.where(np.arange(n) % 2 == 0, 0xFF, 0xF3).astype(np.uint8)
            T[n // 3] = 7
        sa = np.empty(n, np.int32)
        s.Sort(T, sa)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); s.Sort(T, sa); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"n={n:8d} {name:8s} host API median {ts[len(ts)//2]*1e6:9.1f} us  min {ts[0]*1e6:9.1f} us", flush=True)
