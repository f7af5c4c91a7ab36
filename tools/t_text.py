"""Per-kernel-category time of one device-resident sort of a text-like / repetitive input.
usage: t_text.py [enwik|period2|uniform] [MiB]"""
import os, sys, time, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltaq_amd import HipSuffixSort, _abi, workload
from tools import datagen
kind = sys.argv[1] if len(sys.argv) > 1 else "enwik"
n = int(float(sys.argv[2]) * (1 << 20)) if len(sys.argv) > 2 else 256 << 20
if kind == "enwik":
    T = datagen.gen_enwik_like(n, 0xD17A0)
elif kind == "period2":
    T = np.where(np.arange(n) % 2 == 0, 0xFF, 0xF3).astype(np.uint8); T[n // 3] = 7
else:
    T = workload.gen_uniform(n, 5)
L = _abi.load()
s = HipSuffixSort(0)
dT = torch.from_numpy(T).cuda(); dSA = torch.empty(n, dtype=torch.int32, device="cuda")
s.Sort(dT, dSA); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); s.Sort(dT, dSA); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{kind} {n>>20} MiB device-resident: {min(ts)*1e3:.2f} ms  ({n/1e6/min(ts):.0f} MB/s)")
r, a, sm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
L.dq_last_sort_info(ctypes.byref(r), ctypes.byref(a), ctypes.byref(sm))
print(f"rounds {r.value} initial_active {a.value} ({a.value/n:.3f} n) sum_active {sm.value} ({sm.value/n:.2f} n)")
L.dq_profile_enable(1); L.dq_profile_reset()
t0 = time.perf_counter(); s.Sort(dT, dSA); torch.cuda.synchronize(); tp = time.perf_counter() - t0
tot = 0
for c in range(_abi.KERNEL_CATEGORIES):
    ln, ms, el, ab = ctypes.c_int64(), ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
    L.dq_profile_get(c, ctypes.byref(ln), ctypes.byref(ms), ctypes.byref(el), ctypes.byref(ab))
    if ln.value:
        nm = L.dq_profile_kernel_name(c).decode()
        print(f"  {nm:24s} launches {ln.value:4d}  {ms.value:8.3f} ms  {ab.value/1e6/max(ms.value,1e-9):8.0f} GB/s alg")
        tot += ms.value
print(f"  kernel sum {tot:.3f} ms of {tp*1e3:.2f} ms wall (profiled)")
L.dq_profile_enable(0)
