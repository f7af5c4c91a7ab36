#!/usr/bin/env python3
"""Diff.Create on a large pair (default 128 MiB of random bytes, 4000 edits): time, DQ_TRACE timeline of the last call,
raw streams against the oracle's loop.  usage: t_bsdiff_big.py [MiB] [edits]"""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from deltaq_amd import Diff, Patch, _abi
from tools import datagen
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
edits = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rng = np.random.default_rng(3)
old = datagen.gen_uniform(mib << 20, 5)
x = bytearray(old.tobytes())
for _ in range(edits):
    k = int(rng.integers(0, 3)); a = int(rng.integers(0, len(x))); ln = int(rng.integers(1, 400))
    if k == 0: x[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
    elif k == 1: del x[a:a + ln]
    else: x[a:a + ln] = rng.integers(0, 256, min(ln, len(x) - a), dtype=np.uint8).tobytes()
new = np.frombuffer(bytes(x), dtype=np.uint8)
Diff.CreateBytes(old, new)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); patch = Diff.CreateBytes(old, new); ts.append(time.perf_counter() - t0)
print(f"{mib} MiB, {edits} edits: create {min(ts)*1e3:.1f} ms ({', '.join(f'{v*1e3:.0f}' for v in ts)}), patch {len(patch)} bytes, {_abi.last_diff_info()}", flush=True)
os.environ["DQ_TRACE"] = "1"
Diff.CreateBytes(old, new)
del os.environ["DQ_TRACE"]
t0 = time.perf_counter()
ctrl, diff, extra, st = Diff.Scan(old, new)
sa = oracle.divsufsort(old)
wc, wd, we, ns = oracle.bsdiff_scan(old, sa, new)
print("raw streams equal the oracle's:", bool(np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we) and st["searches"] == ns),
      "applies:", Patch.Apply(old, patch) == new.tobytes(), f"(check {time.perf_counter()-t0:.1f} s)", flush=True)
