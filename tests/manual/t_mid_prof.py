#!/usr/bin/env python3
"""Where a mid-size text-like sort spends its time: kernel time per category against the wall clock."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from deltaq_amd import HipSuffixSort, _abi
from tools import datagen

L = _abi.load(); s = HipSuffixSort(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = datagen.gen_enwik_like(n, 0xD17A0, 65536)
dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
for _ in range(3): s.Sort(dT, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): s.Sort(dT, out)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
L.dq_profile_enable(1); L.dq_profile_reset()
s.Sort(dT, out); torch.cuda.synchronize()
L.dq_profile_enable(0)
tot = 0; nl = 0
for k, v in _abi.profile_snapshot().items():
    if v["launches"]:
        print(f"   {k:28s} launches={v['launches']:4d} total={v['ms']*1e3:9.1f} us")
        tot += v["ms"]; nl += v["launches"]
print(f"n={n}: wall {wall*1e6:.0f} us, kernels {tot*1e3:.0f} us in {nl} timed launches, {_abi.last_sort_info()}")
