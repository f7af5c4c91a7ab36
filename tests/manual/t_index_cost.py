#!/usr/bin/env python3
"""Where the time of "one old file, many new files" goes: index build, per-file diffs, teardown, against per-pair Diff.Create."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deltaq_amd import Diff, DiffIndex
from tools import datagen
rng = np.random.default_rng(9)
old = datagen.gen_uniform(16 << 20, 0x5EED0500 + 77)
news = []
for j in range(4):
    x = bytearray(old.tobytes())
    for _ in range(200):
        k, a, ln = int(rng.integers(0, 3)), int(rng.integers(0, len(x))), int(rng.integers(1, 300))
        if k == 0: x[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1: del x[a:a + ln]
        else: x[a:a + ln] = rng.integers(0, 256, min(ln, len(x) - a), dtype=np.uint8).tobytes()
    news.append(np.frombuffer(bytes(x), dtype=np.uint8))
Diff.CreateBytes(old[:4096], news[0][:4096], 0)
for rep in range(2):
    t0 = time.perf_counter(); ix = DiffIndex(old, 0); t1 = time.perf_counter()
    ts = []
    for x in news:
        a = time.perf_counter(); ix.Create(x); ts.append(time.perf_counter() - a)
    t2 = time.perf_counter(); ix.close(); t3 = time.perf_counter()
    ps = []
    for x in news:
        a = time.perf_counter(); Diff.CreateBytes(old, x, 0); ps.append(time.perf_counter() - a)
    print(f"rep {rep}: index build {1e3*(t1-t0):.1f} ms, diffs {[round(1e3*t,1) for t in ts]}, close {1e3*(t3-t2):.1f} ms; per-pair creates {[round(1e3*t,1) for t in ps]}", flush=True)
os.environ["DQ_TRACE"] = "1"
Diff.CreateBytes(old, news[0], 0)
