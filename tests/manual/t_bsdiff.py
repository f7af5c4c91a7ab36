#!/usr/bin/env python3
"""Diff.Create natively (dq_bsdiff_create) on a GPU box against the CPU pipeline it replaces (oracle: LibDivSufSort
restatement + Search / scan loop restatement + libbz2), on file pairs of a few shapes."""
import bz2, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
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

def run(tag, old, new, cpu=True):
    Diff.CreateBytes(old[:1000], new[:1000])
    t0 = time.perf_counter(); patch = Diff.CreateBytes(old, new); gt = time.perf_counter() - t0
    t0 = time.perf_counter(); ctrl, diff, extra, st = Diff.Scan(old, new); st_t = time.perf_counter() - t0
    ok = Patch.Apply(old, patch) == new.tobytes()
    line = f"{tag}: old {old.size} new {new.size}: create {gt*1e3:.1f} ms (scan part {st_t*1e3:.1f} ms: {st}), patch {len(patch)} B, roundtrip {'OK' if ok else 'WRONG'}"
    if cpu:
        t0 = time.perf_counter(); sa = oracle.divsufsort(old); t1 = time.perf_counter()
        wc, wd, we, ns = oracle.bsdiff_scan(old, sa, new); t2 = time.perf_counter()
        z = [bz2.compress(x.tobytes()) for x in (wc, wd, we)]; t3 = time.perf_counter()
        same = np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we)
        line += f"; CPU: sort {1e3*(t1-t0):.0f} + scan {1e3*(t2-t1):.0f} + bzip2 {1e3*(t3-t2):.0f} ms = {1e3*(t3-t0):.0f} ms -> {(t3-t0)/gt:.1f}x; raw streams {'identical' if same else 'DIFFER'}"
    print(line, flush=True)

rng = np.random.default_rng(3)
old = datagen.gen_enwik_like(16 << 20, 3, 64 * 1024)
run("text 16 MiB, 2000 edits", old, edited(rng, old, 2000))
old = datagen.gen_uniform(16 << 20, 5)
run("random 16 MiB, 2000 edits", old, edited(rng, old, 2000))
run("random 16 MiB vs unrelated 4 MiB", old, datagen.gen_uniform(4 << 20, 6))
old = datagen.gen_enwik_like(64 << 20, 7, 64 * 1024)
run("text 64 MiB, 20000 edits", old, edited(rng, old, 20000))
