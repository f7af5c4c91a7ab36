#!/usr/bin/env python3
"""Diff.Create natively under forced settings (window sizes of the scan-loop driver ...): time per pair, windows.
usage: t_bsdiff_variants.py "" "DQ_WIN_MIN=512,DQ_WIN_SECOND=512" ..."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deltaq_amd import Diff, Patch
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

rng = np.random.default_rng(3)
pairs = []
old = datagen.gen_uniform(16 << 20, 5); pairs.append(("random 16 MiB, 2000 edits", old, edited(rng, old, 2000)))
old = datagen.gen_enwik_like(16 << 20, 3, 64 * 1024); pairs.append(("text 16 MiB, 2000 edits", old, edited(rng, old, 2000)))
old = datagen.gen_uniform(16 << 20, 7); pairs.append(("random 16 MiB, 200 edits", old, edited(rng, old, 200)))
old = datagen.gen_uniform(16 << 20, 8); pairs.append(("random 16 MiB, 20000 edits of <= 40 bytes", old, edited(rng, old, 20000, 40)))
pairs.append(("random 16 MiB vs unrelated 4 MiB", datagen.gen_uniform(16 << 20, 5), datagen.gen_uniform(4 << 20, 6)))
variants = sys.argv[1:] or [""]
for name, old, new in pairs:
    first = None
    for v in variants:
        sets = dict(kv.split("=") for kv in v.split(",") if kv)
        for k, val in sets.items(): os.environ[k] = val
        Diff.CreateBytes(old, new)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); patch = Diff.CreateBytes(old, new); ts.append(time.perf_counter() - t0)
        _, _, _, st = Diff.Scan(old, new)
        if first is None: first = patch; ok = Patch.Apply(old, patch) == new.tobytes()
        print(f"{name} [{v or 'defaults'}]: create {min(ts)*1e3:.1f} ms ({', '.join(f'{x*1e3:.0f}' for x in ts)}) {st} same_patch={patch == first} applies={ok}", flush=True)
        for k in sets: del os.environ[k]
