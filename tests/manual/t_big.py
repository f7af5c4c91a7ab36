#!/usr/bin/env python3
"""BASELINE config 4: 2 GiB uniform random, 64-bit SA, checked by sufcheck_i64 + sampled strict order.
Also config 3 timing (256 MiB enwik-like) device-resident with the per-kernel profile."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from deltaq_amd import HipSuffixSort, _abi, workload
from tools import datagen

L = _abi.load(); s = HipSuffixSort(0)
which = sys.argv[1] if len(sys.argv) > 1 else "enwik"
if which == "2g":
    n = 1 << 31
    t0 = time.time(); T = workload.gen_uniform(n, 0x5EED0004); print("gen", round(time.time() - t0, 1), "s", flush=True)
    dT = torch.from_numpy(T).cuda()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    print("workspace GiB", L.dq_sufsort_hip_workspace_bytes(n, 8) / 2**30, flush=True)
    t0 = time.time(); s.Sort(dT, out); torch.cuda.synchronize(); print("sort (first call)", round(time.time() - t0, 3), "s", _abi.last_sort_info(), flush=True)
    L.dq_profile_enable(1); L.dq_profile_reset()
    t0 = time.time(); s.Sort(dT, out); torch.cuda.synchronize(); dt = time.time() - t0
    L.dq_profile_enable(0)
    print(f"2 GiB i64 device-resident: {dt*1e3:.1f} ms = {n/1e6/dt:.0f} MB/s  {_abi.last_sort_info()}", flush=True)
    for k, v in _abi.profile_snapshot().items():
        if v["launches"]:
            print(f"   {k:24s} launches={v['launches']:4d} total={v['ms']:9.3f} ms  alg={v['alg_bytes']/max(v['ms'],1e-9)/1e6:8.1f} GB/s", flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "nocheck":
        sys.exit(0)
    SA = out.cpu().numpy()
    t0 = time.time(); rc = oracle.sufcheck(T, SA); print("sufcheck_i64", rc, round(time.time() - t0, 1), "s", flush=True)
    print("sampled strict order (1e6 pairs):", oracle.verify_sampled(T, SA, 1_000_000, 7), flush=True)
elif which == "i32max":
    # the largest texts the int32 interface takes: 64-bit status words (n >= 2^30) with 32-bit indices,
    # packed words with ib = 31
    for n in ((3 << 29) + 12345, (1 << 31) - 1):
        t0 = time.time(); T = workload.gen_uniform(n, 0x5EED0007); print("gen", n, round(time.time() - t0, 1), "s", flush=True)
        if n == (1 << 31) - 1:
            T[1000:200_000] = T[5_000_000:5_199_000]          # a long repeat: dense rounds at this size too
        dT = torch.from_numpy(T).cuda()
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        t0 = time.time(); s.Sort(dT, out); torch.cuda.synchronize(); print("sort (first call)", round(time.time() - t0, 3), "s", _abi.last_sort_info(), flush=True)
        t0 = time.time(); s.Sort(dT, out); torch.cuda.synchronize(); dt = time.time() - t0
        print(f"n={n} i32 device-resident: {dt*1e3:.1f} ms = {n/1e6/dt:.0f} MB/s", flush=True)
        SA = out.cpu().numpy(); del dT, out
        t0 = time.time(); rc = oracle.sufcheck(T, SA); print("sufcheck", rc, round(time.time() - t0, 1), "s", flush=True)
        print("sampled strict order (1e6 pairs):", oracle.verify_sampled(T, SA, 1_000_000, 7), flush=True)
        del SA, T
elif which == "enwik320":
    # text-like input with bits(n-1) = 29: the suffix-binned ISA build with 32768-entry spans
    n = 320 << 20
    T = datagen.gen_enwik_like(n, 0xD17A1)
    dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
    s.Sort(dT, out); torch.cuda.synchronize()
    t0 = time.time(); s.Sort(dT, out); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"enwik 320 MiB device-resident: {dt*1e3:.1f} ms = {n/1e6/dt:.0f} MB/s  {_abi.last_sort_info()}", flush=True)
    SA = out.cpu().numpy()
    print("sufcheck", oracle.sufcheck(T, SA), "sampled", oracle.verify_sampled(T, SA, 1_000_000, 3), flush=True)
else:
    n = 256 << 20
    T = datagen.gen_enwik_like(n, 0xD17A0)
    dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
    s.Sort(dT, out); torch.cuda.synchronize()
    L.dq_profile_enable(1); L.dq_profile_reset()
    t0 = time.time(); s.Sort(dT, out); torch.cuda.synchronize(); dt = time.time() - t0
    L.dq_profile_enable(0)
    print(f"enwik 256 MiB device-resident: {dt*1e3:.1f} ms = {n/1e6/dt:.0f} MB/s  {_abi.last_sort_info()}")
    for k, v in _abi.profile_snapshot().items():
        if v["launches"]:
            print(f"   {k:24s} launches={v['launches']:4d} total={v['ms']:9.3f} ms  alg={v['alg_bytes']/max(v['ms'],1e-9)/1e6:8.1f} GB/s")
    SA = out.cpu().numpy()
    print("sufcheck", oracle.sufcheck(T, SA), "sampled", oracle.verify_sampled(T, SA, 1_000_000, 3))
