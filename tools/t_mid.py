"""Host-API latency on text-like inputs of bsdiff-typical sizes (64 KiB .. 16 MiB)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaq_amd import HipSuffixSort, _abi
from tools import datagen
s = HipSuffixSort(0)
for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
    T = datagen.gen_enwik_like(n, 0xD17A0, min(n // 4, 64 * 1024))
    sa = np.empty(n, np.int32)
    s.Sort(T, sa)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); s.Sort(T, sa); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"n={n:9d} enwik-like host API median {ts[3]*1e3:8.3f} ms  ({n/1e6/ts[3]:7.0f} MB/s)  {_abi.last_sort_info()}", flush=True)
