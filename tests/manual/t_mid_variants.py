#!/usr/bin/env python3
"""Mid-size text-like inputs under forced settings: wall clock per sort (device-resident), launches, rounds.
usage: t_mid_variants.py "" "DQ_MID_SHORT=1" ..."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from deltaq_amd import HipSuffixSort, _abi
from tools import datagen

L = _abi.load(); s = HipSuffixSort(0)
variants = sys.argv[1:] or [""]
for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24):
    T = datagen.gen_enwik_like(n, 0xD17A0, 65536)
    dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
    first = None
    for v in variants:
        sets = dict(kv.split("=") for kv in v.split(",") if kv)
        for k, val in sets.items(): os.environ[k] = val
        for _ in range(3): s.Sort(dT, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): s.Sort(dT, out)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
        L.dq_profile_enable(1); L.dq_profile_reset()
        s.Sort(dT, out); torch.cuda.synchronize()
        L.dq_profile_enable(0)
        snap = _abi.profile_snapshot()
        nl = sum(p["launches"] for p in snap.values()); kt = sum(p["ms"] for p in snap.values())
        same = True
        if first is None: first = out.clone()
        else: same = bool(torch.equal(first, out))
        print(f"n={n:9d} [{v or 'defaults':28s}] {wall*1e6:8.1f} us  kernels {kt*1e3:7.0f} us in {nl:3d} timed launches  {_abi.last_sort_info()}  same={same}", flush=True)
        if os.environ.get("T_DETAIL"):
            for k, p in snap.items():
                if p["launches"]: print(f"      {k:28s} launches={p['launches']:4d} total={p['ms']*1e3:9.1f} us")
        for k in sets: del os.environ[k]
