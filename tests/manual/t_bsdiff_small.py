#!/usr/bin/env python3
"""Diff.Create on small pairs (256 KiB .. 4 MiB, one edit per ~8 KiB) under forced settings.
usage: t_bsdiff_small.py "" "DQ_SCAN_MIN_SEG=262144" ..."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deltaq_amd import Diff, _abi
from tools import datagen
rng = np.random.default_rng(4)
def edited(old, edits, span=300):
    x = bytearray(old.tobytes())
    for _ in range(edits):
        k, a, ln = int(rng.integers(0, 3)), int(rng.integers(0, len(x))), int(rng.integers(1, span))
        if k == 0: x[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1: del x[a:a + ln]
        else: x[a:a + ln] = rng.integers(0, 256, min(ln, len(x) - a), dtype=np.uint8).tobytes()
    return np.frombuffer(bytes(x), dtype=np.uint8)
pairs = []
for kib in (256, 1024, 2048, 4096):
    old = datagen.gen_uniform(kib << 10, 11 + kib); pairs.append((f"random {kib} KiB", old, edited(old, kib // 8)))
    old = datagen.gen_enwik_like(kib << 10, 5 + kib, 16384); pairs.append((f"text {kib} KiB", old, edited(old, kib // 8)))
for name, old, new in pairs:
    first = None
    for v in (sys.argv[1:] or [""]):
        sets = dict(kv.split("=") for kv in v.split(",") if kv)
        for k, val in sets.items(): os.environ[k] = val
        Diff.CreateBytes(old, new)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); p = Diff.CreateBytes(old, new); ts.append(time.perf_counter() - t0)
        info = _abi.last_diff_info()
        if first is None: first = p
        print(f"{name} [{v or 'defaults'}]: create {min(ts)*1e3:.2f} ms (median {sorted(ts)[3]*1e3:.2f}) chains {info['chains_launched']} joined {info['chains_joined']} same_patch={p == first}", flush=True)
        for k in sets: del os.environ[k]
