#!/usr/bin/env python3
"""Real data from the image itself (not synthetic, not the reference): a shared library (machine code + tables),
Python sources (text), a tar-like mix.  Each: device-resident sort time, the path taken (DQ_TRACE), sufcheck +
sampled strict order; smaller prefixes are bit-compared with the oracle."""
import glob, os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from deltaq_amd import HipSuffixSort, _abi

def read_prefix(path, nbytes, skip=0):
    with open(path, "rb") as f:
        f.seek(skip)
        return np.frombuffer(f.read(nbytes), dtype=np.uint8)

def python_sources(nbytes):
    out, tot = [], 0
    for p in sorted(glob.glob("/usr/lib/python3*/**/*.py", recursive=True)):
        try:
            b = open(p, "rb").read()
        except OSError:
            continue
        out.append(b); tot += len(b)
        if tot >= nbytes: break
    return np.frombuffer(b"".join(out)[:nbytes], dtype=np.uint8)

s = HipSuffixSort(0)
libs = sorted(glob.glob("/opt/rocm/lib/librocsparse.so.*"), key=os.path.getsize)
cases = []
if libs:
    cases.append(("librocsparse.so, 64 MiB from offset 0", read_prefix(libs[-1], 64 << 20)))
    cases.append(("librocsparse.so, 256 MiB from offset 64 MiB", read_prefix(libs[-1], 256 << 20, 64 << 20)))
src = python_sources(48 << 20)
cases.append((f"python sources, {src.size >> 20} MiB", src))
if libs:
    cases.append(("mix: sources + library + sources", np.concatenate([src[: 8 << 20], read_prefix(libs[-1], 16 << 20, 1 << 20), src[: 8 << 20]])))
for name, T in cases:
    T = np.ascontiguousarray(T)
    n = T.size
    cnt = np.bincount(T, minlength=256); p = cnt[cnt > 0] / n
    h0 = float(-(p * np.log2(p)).sum())
    dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
    os.environ["DQ_TRACE"] = "1"
    s.Sort(dT, out); torch.cuda.synchronize()
    os.environ.pop("DQ_TRACE")
    t0 = time.perf_counter()
    for _ in range(3): s.Sort(dT, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    SA = out.cpu().numpy()
    ok = oracle.sufcheck(T, SA)
    samp = oracle.verify_sampled(T, SA, 200_000, 5)
    exact = ""
    if n <= (64 << 20):
        t0 = time.perf_counter(); ref = oracle.divsufsort(T); ct = time.perf_counter() - t0
        exact = f", == oracle: {bool(np.array_equal(ref, SA))} (CPU {ct:.1f} s = {ct/dt:.0f}x)"
    print(f"{name}: n={n} sigma={int((cnt>0).sum())} H0={h0:.2f}: {dt*1e3:.1f} ms = {n/1e6/dt:.0f} MB/s {_abi.last_sort_info()} sufcheck={ok} sampled={samp}{exact}", flush=True)
