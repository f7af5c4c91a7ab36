#!/usr/bin/env python3
"""Mid-size text-like inputs (64 KiB ... 16 MiB): device-resident and host-interface times, rounds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from deltaq_amd import HipSuffixSort, _abi
from tools import datagen

L = _abi.load(); s = HipSuffixSort(0)
for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
    for kind in ("enwik", "uniform"):
        T = datagen.gen_enwik_like(n, 0xD17A0, 65536) if kind == "enwik" else datagen.gen_uniform(n, 5)
        dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
        for _ in range(3): s.Sort(dT, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): s.Sort(dT, out)
        torch.cuda.synchronize(); dev = (time.perf_counter() - t0) / 10
        sa = np.empty(n, np.int32)
        s.Sort(T, sa)
        t0 = time.perf_counter()
        for _ in range(5): s.Sort(T, sa)
        host = (time.perf_counter() - t0) / 5
        print(f"{kind:8s} n={n:9d}  device {dev*1e6:8.1f} us   host interface {host*1e6:8.1f} us   {_abi.last_sort_info()}", flush=True)
