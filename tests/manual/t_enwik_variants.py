#!/usr/bin/env python3
"""256 MiB enwik-style text (BASELINE configs[2]) under several forced settings: time, per-kernel profile, rounds;
every variant's suffix array is compared with the first one's on the device, the first one checked by sufcheck.
usage: t_enwik_variants.py "DQ_MID_GROUPS=0" "DQ_MID_GROUPS=256" ...   ("" = defaults; several settings: "A=1,B=2")"""
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
n = int(os.environ.get("T_MIB", "256")) << 20
T = datagen.gen_enwik_like(n, 0xD17A0)
dT = torch.from_numpy(T).cuda()
first = None
variants = sys.argv[1:] or [""]
for v in variants:
    sets = dict(kv.split("=") for kv in v.split(",") if kv)
    for k, val in sets.items(): os.environ[k] = val
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    s.Sort(dT, out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); s.Sort(dT, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    L.dq_profile_enable(1); L.dq_profile_reset()
    s.Sort(dT, out); torch.cuda.synchronize()
    L.dq_profile_enable(0)
    print(f"== [{v or 'defaults'}] {min(ts)*1e3:.2f} ms (min of 3; {', '.join(f'{x*1e3:.2f}' for x in ts)})  {_abi.last_sort_info()}", flush=True)
    for k, p in _abi.profile_snapshot().items():
        if p["launches"]:
            print(f"   {k:26s} launches={p['launches']:4d} total={p['ms']:8.3f} ms  alg={p['alg_bytes']/max(p['ms'],1e-9)/1e6:8.1f} GB/s", flush=True)
    if os.environ.get("T_TRACE"):
        os.environ["DQ_TRACE"] = "1"; s.Sort(dT, out); torch.cuda.synchronize(); del os.environ["DQ_TRACE"]
    if first is None:
        first = out
        SA = out.cpu().numpy()
        print("   sufcheck", oracle.sufcheck_mt(T, SA) if hasattr(oracle, "sufcheck_mt") else oracle.sufcheck(T, SA),
              "sampled", oracle.verify_sampled(T, SA, 1_000_000, 3), flush=True)
    else:
        print("   equals the first variant's SA:", bool(torch.equal(first, out)), flush=True)
    for k in sets: del os.environ[k]
