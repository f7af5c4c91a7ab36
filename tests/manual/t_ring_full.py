#!/usr/bin/env python3
"""A pair whose scan writes more list entries than a chain's list holds per launch (32 768): 16 MiB, one byte changed every
150 bytes -- the chain leaves with its list full, is read to the end and launched again from where it stood.
Raw streams and Search count against the oracle's loop, with 1 / 2 / 16 grids."""
import os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle
from deltaq_amd import Diff, _abi
from tools import datagen
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
old = datagen.gen_uniform(mib << 20, 91)
new = old.copy()
new[100::150] ^= 0x5a
sa = oracle.divsufsort(old)
wc, wd, we, ns = oracle.bsdiff_scan(old, sa, new)
print("oracle:", np.asarray(wc).size // 3, "triples,", ns, "searches", flush=True)
for env in ({"DQ_SCAN_CHAINS": "1"}, {"DQ_SCAN_CHAINS": "2"}, {}):
    for k, v in env.items(): os.environ[k] = v
    t0 = time.perf_counter()
    ctrl, diff, extra, st = Diff.Scan(old, new)
    dt = time.perf_counter() - t0
    info = _abi.last_diff_info()
    ok = np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we) and st["searches"] == ns
    print(env or "defaults", "equal the oracle's:", bool(ok), f"{dt*1e3:.1f} ms", info, flush=True)
    for k in env: del os.environ[k]
