#!/usr/bin/env python3
"""Runs every BASELINE.json config (+ the 256 MiB target run) on one GPU box and prints the
results table of BASELINE.md section 4.  CPU numbers: oracle/divsufsort.c, 1 thread, same host."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from deltaq_amd import HipSuffixSort, _abi, workload
from tools import datagen

L = _abi.load(); s = HipSuffixSort(0)
rows = []

def gpu_times(T, dtype, reps=5):
    n = T.size
    tdt = torch.int32 if dtype == np.int32 else torch.int64
    dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=tdt, device="cuda")
    s.Sort(dT, out); torch.cuda.synchronize()
    L.dq_profile_reset(); L.dq_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps): s.Sort(dT, out)
    torch.cuda.synchronize(); dev = (time.perf_counter() - t0) / reps
    L.dq_profile_enable(0)
    rs = _abi.profile_snapshot()["radix_rank_kernel"]
    frac = rs["alg_bytes"] / (rs["ms"] * 1e-3) / 8e12 if rs["launches"] else 0.0
    sa = np.empty(n, dtype=dtype)
    s.Sort(T, sa)                                   # warm host path (workspace growth)
    t0 = time.perf_counter(); s.Sort(T, sa); host = time.perf_counter() - t0
    gsa = out.cpu().numpy()
    assert np.array_equal(gsa, sa)
    del dT, out
    return dev, host, frac, sa

def run(name, T, dtype=np.int32, cpu=True, full_compare=True):
    n = T.size
    dev, host, frac, sa = gpu_times(T, dtype)
    info = _abi.last_sort_info()
    cpu_mbs = None; exact = None
    if cpu:
        t0 = time.perf_counter(); ref = oracle.divsufsort(T, dtype=dtype); ct = time.perf_counter() - t0
        cpu_mbs = n / 1e6 / ct
        exact = bool(np.array_equal(ref, sa))
    else:
        exact = (oracle.sufcheck(T, sa) == 0 and oracle.verify_sampled(T, sa, 1_000_000, 9) == -1)
    row = dict(config=name, n=n, cpu_MBps=cpu_mbs, gpu_dev_MBps=n / 1e6 / dev, gpu_abi_MBps=n / 1e6 / host,
               speedup_dev=(n / 1e6 / dev) / cpu_mbs if cpu_mbs else None,
               speedup_abi=(n / 1e6 / host) / cpu_mbs if cpu_mbs else None,
               rank_frac=frac, rounds=info["rounds"], bit_exact=exact, dev_ms=dev * 1e3, abi_ms=host * 1e3)
    rows.append(row); print(json.dumps(row), flush=True)

# 1: plumbing, CPU reference path only
T = oracle.net_random_bytes(4096)
sa = oracle.divsufsort(T); import hashlib
print("config 1: oracle SA sha256", hashlib.sha256(sa.astype("<i4").tobytes()).hexdigest(), "sufcheck", oracle.sufcheck(T, sa), flush=True)
run("2: 64 MiB uniform", workload.gen_uniform(64 << 20, 0x5EED0002))
run("T: 256 MiB uniform", workload.gen_uniform(256 << 20, 0x5EED0003))
run("3: 256 MiB enwik-like", datagen.gen_enwik_like(256 << 20, 0xD17A0))
if "--2g" in sys.argv:
    run("4: 2 GiB uniform, i64", workload.gen_uniform(1 << 31, 0x5EED0004), dtype=np.int64, cpu="--2g-cpu" in sys.argv)
# 5: 128 x 16 MiB on ONE GPU (batch entry point, host buffers) + device-resident loop
texts = [workload.gen_uniform(16 << 20, 0x5EED0500 + j) for j in range(128)]
sas = [np.ones(t.size, np.int32) for t in texts]           # pre-touched output pages (np.zeros would stay untouched)
cnt = len(texts)
tp = (ctypes.c_void_p * cnt)(*[t.ctypes.data for t in texts]); sp = (ctypes.c_void_p * cnt)(*[a.ctypes.data for a in sas])
ln = (ctypes.c_int64 * cnt)(*[t.size for t in texts])
t0 = time.perf_counter(); rc = L.dq_sufsort_hip_batch_i32(cnt, tp, ln, sp, 1, None); bt = time.perf_counter() - t0
assert rc == 0
dT = [torch.from_numpy(t).cuda() for t in texts[:16]]; out = torch.empty(16 << 20, dtype=torch.int32, device="cuda")
s.Sort(dT[0], out); torch.cuda.synchronize(); t0 = time.perf_counter()
for d in dT: s.Sort(d, out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 16
t0 = time.perf_counter(); ref = oracle.divsufsort(texts[0]); ct = time.perf_counter() - t0
ok = all(oracle.sufcheck(t, a) == 0 for t, a in list(zip(texts, sas))[:8]) and np.array_equal(ref, sas[0])
row = dict(config="5: 128 x 16 MiB @1 GPU", n=128 * (16 << 20), cpu_MBps=(16 << 20) / 1e6 / ct, gpu_dev_MBps=(16 << 20) / 1e6 / dt,
           gpu_abi_MBps=128 * (16 << 20) / 1e6 / bt, bit_exact=bool(ok))
rows.append(row); print(json.dumps(row), flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
