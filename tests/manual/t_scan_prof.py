#!/usr/bin/env python3
"""Scan loop of Diff.Create between similar files: wall time against the match-search kernel time and launches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deltaq_amd import Diff, _abi
from tools import datagen

def edited(rng, old, edits, span=400):
    new = bytearray(old.tobytes())
    for _ in range(edits):
        k = int(rng.integers(0, 4)); a = int(rng.integers(0, max(1, len(new)))); ln = int(rng.integers(1, span))
        if k == 0: new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1: del new[a:a + ln]
        elif k == 2: new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
        else: new[a:a] = new[max(0, a - 3 * ln):max(0, a - 2 * ln)]
    return np.frombuffer(bytes(new), dtype=np.uint8)

L = _abi.load()
rng = np.random.default_rng(3)
for tag, old in (("text", datagen.gen_enwik_like(16 << 20, 3, 64 * 1024)), ("random", datagen.gen_uniform(16 << 20, 5))):
    new = edited(rng, old, 2000)
    Diff.Scan(old[:100000], new[:100000])
    L.dq_profile_enable(100 + 16); L.dq_profile_reset()
    t0 = time.perf_counter(); ctrl, diff, extra, st = Diff.Scan(old, new); dt = time.perf_counter() - t0
    L.dq_profile_enable(0)
    ms = _abi.profile_snapshot()["match_search_kernel"]
    print(f"{tag}: scan {dt*1e3:.1f} ms, {st}; match_search_kernel: {ms['launches']} launches, {ms['ms']:.1f} ms total = {ms['ms']*1e3/max(ms['launches'],1):.1f} us each")
