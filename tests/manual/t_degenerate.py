#!/usr/bin/env python3
"""Inputs with long repeats (the reference bounds such work with its Budget, TrSort.cs:30-101): device-resident time of
each, checked by sufcheck + sampled strict pairs.  usage: t_degenerate.py [MiB]"""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from deltaq_amd import HipSuffixSort, _abi
from tools import datagen

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = mib << 20
rnd = datagen.gen_uniform(n, 77)
def fib(n):
    a, b = b"b", b"a"
    while len(b) < n: a, b = b, b + a
    return np.frombuffer(b[:n], np.uint8).copy()
cases = {
    "zeros": np.zeros(n, np.uint8),
    "period 3": np.tile(np.frombuffer(b"abc", np.uint8), n // 3 + 1)[:n].copy(),
    "period 65": np.tile(rnd[:65], n // 65 + 1)[:n].copy(),
    "period 1000": np.tile(rnd[:1000], n // 1000 + 1)[:n].copy(),
    "period 1 MiB": np.tile(rnd[:1 << 20], mib)[:n].copy(),
    "two copies of n/2 random bytes": np.concatenate([rnd[:n // 2], rnd[:n // 2]]),
    "Fibonacci word": fib(n),
    "random, 4 symbols": (rnd & 3).copy(),
    "random bytes with 1000 copies of one 10 KB block": None,
}
x = rnd.copy()
rng = np.random.default_rng(5)
for a in rng.integers(0, n - 10240, 1000): x[a:a + 10240] = rnd[:10240]
cases["random bytes with 1000 copies of one 10 KB block"] = x
s = HipSuffixSort(0)
for name, T in cases.items():
    T = np.ascontiguousarray(T)
    dT = torch.from_numpy(T).cuda()
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    s.Sort(dT, out); torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); s.Sort(dT, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    SA = out.cpu().numpy()
    ok = oracle.verify_sampled(T, SA, 200_000, 3)
    print(f"{mib} MiB {name}: {min(ts)*1e3:.2f} ms  {_abi.last_sort_info()}  sampled strict pairs first bad: {ok}", flush=True)
