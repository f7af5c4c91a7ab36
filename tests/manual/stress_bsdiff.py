#!/usr/bin/env python3
"""Randomised differential run of the native Diff.Create / Patch.Apply on a GPU box (not collected by pytest;
STRESS_LOG=<file> keeps the pair being worked on, for the post-mortem of a crash or a hang):
    python tests/manual/stress_bsdiff.py [seconds] [seed]
Random old files of 0 .. 3 MB (uniform, few symbols, text-like, periodic), new = old with random edits / an unrelated
file / a prefix; raw streams compared with the oracle's scan loop, patches applied back."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from deltaq_amd import Diff, Patch
from tools import datagen

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)

def make_old(n):
    k = int(rng.integers(0, 5))
    if k == 0: return datagen.gen_uniform(n, int(rng.integers(1, 1 << 30)))
    if k == 1: return datagen.gen_uniform(n, int(rng.integers(1, 1 << 30))) & int(rng.integers(1, 16))
    if k == 2: return datagen.gen_enwik_like(n, int(rng.integers(1, 1 << 30)), int(rng.integers(256, 1 << 16)))
    if k == 3: return np.tile(datagen.gen_uniform(int(rng.integers(64, 5000)), 7), n // 64 + 1)[:n].copy()
    # one symbol: the reference's own loop is quadratic between such files (a 1.3 MB pair with one 6-byte insertion
    # takes its restatement 10 minutes: 930 000 searches of ~1 MB each), so they stay short here
    return np.zeros(min(n, 30_000), np.uint8)

def edited(old):
    new = bytearray(old.tobytes())
    for _ in range(int(rng.integers(0, 40))):
        k = int(rng.integers(0, 4)); a = int(rng.integers(0, max(1, len(new)))); ln = int(rng.integers(1, 3000))
        if k == 0: new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1: del new[a:a + ln]
        elif k == 2: new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
        else: new[a:a] = new[max(0, a - 3 * ln):max(0, a - 2 * ln)]
    return np.frombuffer(bytes(new), dtype=np.uint8)

t_end = time.time() + budget
count = 0
while time.time() < t_end:
    u = rng.random()
    n = int(rng.integers(0, 300)) if u < 0.15 else int(rng.integers(300, 100_000)) if u < 0.7 else int(rng.integers(100_000, 3_000_000))
    old = np.ascontiguousarray(make_old(n), dtype=np.uint8)
    v = rng.random()
    new = edited(old) if v < 0.7 else (make_old(int(rng.integers(0, 200_000))) if v < 0.85 else old[: int(rng.integers(0, n + 1))].copy())
    new = np.ascontiguousarray(new, dtype=np.uint8)
    if os.environ.get("STRESS_LOG"):
        with open(os.environ["STRESS_LOG"], "w") as f:
            f.write(repr(dict(count=count, n=n, m=int(new.size), t=round(time.time() - (t_end - budget), 1))) + "\n")
        np.save(os.environ["STRESS_LOG"] + ".old.npy", old); np.save(os.environ["STRESS_LOG"] + ".new.npy", new)
    # (the device's anchor search with grids of several sizes, and the host loop over windows it falls back to)
    # (several grids on one file -- "chains", dq_diff.hip -- start at 256 KiB of new by default: forced onto these short
    # files with segments from 64 bytes on, 2 .. 8 grids, 1 .. 8 iteration ends walked into the next grid's part)
    for k in ("DQ_SCAN_DEVICE", "DQ_SCAN_GROUPS", "DQ_SCAN_CHAINS", "DQ_SCAN_MIN_SEG", "DQ_SCAN_EXTRA", "DQ_SCAN_LANE_BUDGET", "DQ_SCAN_PAR_EMIT"): os.environ.pop(k, None)
    mode = int(rng.integers(0, 10))
    if mode == 0: os.environ["DQ_SCAN_DEVICE"] = "0"
    elif mode == 1: os.environ["DQ_SCAN_GROUPS"] = "8"
    elif mode == 2: os.environ["DQ_SCAN_GROUPS"] = "48"
    elif mode >= 5:
        os.environ["DQ_SCAN_CHAINS"] = str(int(rng.integers(2, 9)))
        os.environ["DQ_SCAN_MIN_SEG"] = str(int(rng.choice([64, 500, 4096, 30_000, 200_000])))
        os.environ["DQ_SCAN_GROUPS"] = str(int(rng.choice([8, 16, 32])))
        os.environ["DQ_SCAN_EXTRA"] = str(int(rng.choice([1, 2, 8])))
        os.environ["DQ_SCAN_LANE_BUDGET"] = str(int(rng.choice([1, 2, 4])))
        if rng.random() < 0.25: os.environ["DQ_SCAN_PAR_EMIT"] = "0"       # (no emitter threads of the chains' own)
    t0 = time.time(); ctrl, diff, extra, st = Diff.Scan(old, new); t1 = time.time()
    sa = oracle.divsufsort(old)
    wc, wd, we, ns = oracle.bsdiff_scan(old, sa, new); t2 = time.time()
    if t2 - t0 > 5.0:
        print(f"slow pair {count}: n={n} m={new.size} device {t1-t0:.1f} s, oracle {t2-t1:.1f} s, {st}, old[:8]={old[:8].tolist()}", flush=True)
    ok = np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we) and st["searches"] == ns
    what = "raw streams / Search count"
    if ok and count % 4 == 0:
        # framing: behind the running scan from the first byte / behind it by default (files >= 256 KiB) / after it;
        # block transform with and without the shortcuts for doubled blocks and run-length coded diff streams
        for k in ("DQ_FRAME_FOLLOW_MIN", "DQ_FRAME_AFTER", "DQ_NO_TWINS", "DQ_NO_PERIOD_HINT"): os.environ.pop(k, None)
        fm = int(rng.integers(0, 4))
        if fm == 0: os.environ["DQ_FRAME_FOLLOW_MIN"] = "0"
        elif fm == 1: os.environ["DQ_FRAME_AFTER"] = "1"
        elif fm == 2: os.environ["DQ_FRAME_FOLLOW_MIN"] = "0"; os.environ["DQ_NO_TWINS"] = "1"; os.environ["DQ_NO_PERIOD_HINT"] = "1"
        patch = Diff.CreateBytes(old, new)
        ok = Patch.Apply(old, patch) == new.tobytes()
        what = "round trip"
        if ok:                                            # ... and libbz2 reads the three streams as the raw streams they are
            import bz2, struct
            def plong(b):
                v = struct.unpack("<Q", b)[0]
                return -(v & ~(1 << 63)) if v >> 63 else v
            lc, ld = plong(patch[8:16]), plong(patch[16:24])
            z = [patch[32:32 + lc], patch[32 + lc:32 + lc + ld], patch[32 + lc + ld:]]
            got = [bz2.decompress(x) if len(x) else b"" for x in z]
            ok = (got[1] == wd.tobytes() and got[2] == we.tobytes() and len(got[0]) == 8 * wc.size and
                  [plong(got[0][i:i + 8]) for i in range(0, len(got[0]), 8)] == wc.ravel().tolist())
            what = "streams as libbz2 reads them"
    if not ok:
        print("failed:", what, "stats", st, "oracle searches", ns, {k: v for k, v in os.environ.items() if k.startswith("DQ_")}, flush=True)
        for rep in range(3):                              # the same pair again, in this process
            c2, d2, e2, st2 = Diff.Scan(old, new)
            p2 = Diff.CreateBytes(old, new)
            print("  again: raw streams", np.array_equal(c2, wc) and np.array_equal(d2, wd) and np.array_equal(e2, we),
                  "round trip", Patch.Apply(old, p2) == new.tobytes(), "same patch bytes", p2 == (patch if what == "round trip" else p2), flush=True)
        np.save(os.path.join(ROOT, "gpurun_out", f"bsdiff_fail_old_{seed}_{count}.npy"), old)
        np.save(os.path.join(ROOT, "gpurun_out", f"bsdiff_fail_new_{seed}_{count}.npy"), new)
        print("MISMATCH", dict(count=count, n=n, m=new.size), flush=True)
        sys.exit(1)
    count += 1
print(f"bsdiff stress OK: {count} file pairs, seed {seed}", flush=True)
