#!/usr/bin/env python3
"""One old file, several new files: where the time of the DiffIndex form goes (build, each Create, release) beside
Diff.CreateBytes per pair.  usage: t_index_many.py"""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deltaq_amd import Diff, DiffIndex
from tools import datagen
rng = np.random.default_rng(9)
old = datagen.gen_uniform(16 << 20, 77)
news = []
for j in range(4):
    x = bytearray(old.tobytes())
    for _ in range(200):
        k, a, ln = int(rng.integers(0, 3)), int(rng.integers(0, len(x))), int(rng.integers(1, 300))
        if k == 0: x[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1: del x[a:a + ln]
        else: x[a:a + ln] = rng.integers(0, 256, min(ln, len(x) - a), dtype=np.uint8).tobytes()
    news.append(np.frombuffer(bytes(x), dtype=np.uint8))
Diff.CreateBytes(old, news[0], 0)
for rep in range(3):
    t0 = time.perf_counter()
    ix = DiffIndex(old, 0); ix.__enter__()
    t1 = time.perf_counter()
    ts = []
    for x in news:
        a = time.perf_counter(); p = ix.Create(x); ts.append(time.perf_counter() - a)
    t2 = time.perf_counter()
    ix.__exit__(None, None, None)
    t3 = time.perf_counter()
    tp = []
    for x in news:
        a = time.perf_counter(); q = Diff.CreateBytes(old, x, 0); tp.append(time.perf_counter() - a)
    print(f"index build {1e3*(t1-t0):.2f} ms, Create {[round(1e3*v,2) for v in ts]} ms, release {1e3*(t3-t2):.2f} ms; per-pair CreateBytes {[round(1e3*v,2) for v in tp]} ms", flush=True)
