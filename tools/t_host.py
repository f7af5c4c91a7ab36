import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaq_amd import HipSuffixSort, workload
s = HipSuffixSort(0)
for mib in (1, 16, 64, 256):
    n = mib << 20
    T = workload.gen_uniform(n, 5); sa = np.empty(n, np.int32)
    s.Sort(T, sa)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); s.Sort(T, sa); ts.append(time.perf_counter() - t0)
    print(f"{mib:4d} MiB host API: {min(ts)*1e3:8.2f} ms  ({n/1e6/min(ts):8.0f} MB/s of text)", flush=True)
