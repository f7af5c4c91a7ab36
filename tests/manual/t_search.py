#!/usr/bin/env python3
"""Match search (dq_bsdiff_search_dev_i32) timing on a GPU box against the oracle's Search restatement (1 CPU thread):
16 MiB / 64 MiB old, 10^6 / 4*10^6 scan positions in a differing region (two independent random buffers) and in
text-like data with edits."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from deltaq_amd import HipSuffixSort, HipMatchSearch, _abi
from tools import datagen

L = _abi.load(); ss = HipSuffixSort(0); ms = HipMatchSearch(0)

def run(tag, old, new, count, cpu_count):
    d_old = torch.from_numpy(old).cuda(); d_new = torch.from_numpy(new).cuda()
    d_sa = ss.Sort(d_old)
    ms.Search(d_sa, d_old, d_new, scan0=0, count=1000); torch.cuda.synchronize()
    L.dq_profile_reset(); L.dq_profile_enable(1)
    t0 = time.perf_counter()
    pos, ln = ms.Search(d_sa, d_old, d_new, scan0=0, count=count)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    L.dq_profile_enable(0)
    k = _abi.profile_snapshot()["match_search_kernel"]
    sa = d_sa.cpu().numpy()
    t0 = time.perf_counter(); w = oracle.bsdiff_search(old, sa, new, scan0=0, count=cpu_count); ct = time.perf_counter() - t0
    ok = np.array_equal(ln.cpu().numpy()[:cpu_count], w[1]) and np.array_equal(pos.cpu().numpy()[:cpu_count], w[0])
    print(f"{tag}: n={old.size} queries={count}: kernel {k['ms']:.3f} ms = {count / k['ms'] / 1e3:.1f} M queries/s (wall {dt*1e3:.2f} ms); "
          f"CPU {cpu_count / ct / 1e6:.3f} M queries/s -> {count / k['ms'] / 1e3 / (cpu_count / ct / 1e6):.0f}x; mean len {float(w[1].mean()):.1f}; {'OK' if ok else 'WRONG'}", flush=True)

rnd = datagen.gen_uniform
run("random vs random 16 MiB", rnd(16 << 20, 1), rnd(16 << 20, 2), 1_000_000, 200_000)
run("random vs random 64 MiB", rnd(64 << 20, 1), rnd(64 << 20, 2), 4_000_000, 200_000)
old = datagen.gen_enwik_like(16 << 20, 3, 64 * 1024)
new = old.copy(); rng = np.random.default_rng(1)
for a in rng.integers(0, new.size - 100, 20000): new[a:a + 8] = rng.integers(0, 256, 8, dtype=np.uint8)
run("text with 20000 edits 16 MiB", old, new, 1_000_000, 100_000)
