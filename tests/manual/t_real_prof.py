#!/usr/bin/env python3
"""Per-kernel profile of one real-data sort (a 64 MiB slice of a shared library of the image)."""
import glob, os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from deltaq_amd import HipSuffixSort, _abi
L = _abi.load(); s = HipSuffixSort(0)
lib = sorted(glob.glob("/opt/rocm/lib/librocsparse.so.*"), key=os.path.getsize)[-1]
T = np.fromfile(lib, dtype=np.uint8, count=64 << 20)
dT = torch.from_numpy(T).cuda(); out = torch.empty(T.size, dtype=torch.int32, device="cuda")
s.Sort(dT, out); torch.cuda.synchronize()
t0 = time.perf_counter(); s.Sort(dT, out); torch.cuda.synchronize(); wall = time.perf_counter() - t0
L.dq_profile_enable(1); L.dq_profile_reset()
os.environ["DQ_TRACE"] = "1"
s.Sort(dT, out); torch.cuda.synchronize()
L.dq_profile_enable(0)
tot = 0
for k, v in _abi.profile_snapshot().items():
    if v["launches"]:
        print(f"   {k:28s} launches={v['launches']:4d} total={v['ms']:8.3f} ms"); tot += v["ms"]
print(f"wall {wall*1e3:.2f} ms, kernels {tot:.2f} ms, {_abi.last_sort_info()}")
