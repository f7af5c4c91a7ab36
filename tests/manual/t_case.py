#!/usr/bin/env python3
"""One named input under several forced settings: device-resident time, per-kernel profile, DQ_TRACE of the rounds;
every variant's suffix array is compared with the first one's (which is checked by sufcheck + sampled strict pairs).
usage: t_case.py <uniform256|enwik256|enwik64|textk1024|libtorch128|rocsparse256|rocsparse64> "" "DQ_X=1,DQ_Y=2" ...   (T_TRACE=1: per-round trace)"""
import glob, os, sys, time
os.environ.setdefault("DQ_DEBUG_FLAGS", "1")      # the library honours its DQ_* overrides only under this gate
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import oracle
from deltaq_amd import HipSuffixSort, _abi
from tools import datagen


def load_case(name):
    if name.startswith("textk"):                          # text of <k> KiB (the reference benchmark's size range)
        return datagen.gen_enwik_like(int(name[5:]) << 10, 0xD17A0, 64 * 1024)
    if name.startswith("uniform"):                        # the north-star buffer (256) and its smaller siblings
        return datagen.gen_uniform(int(name[7:]) << 20, 0x5EED0003)
    if name.startswith("enwik"):
        return datagen.gen_enwik_like(int(name[5:]) << 20, 0xD17A0)
    if name.startswith("libtorch"):
        p = sorted(glob.glob("/usr/local/lib/python3*/dist-packages/torch/lib/libtorch_cpu.so"))[0]
        return np.fromfile(p, dtype=np.uint8, count=int(name[8:]) << 20)
    if name.startswith("rocsparse"):
        p = sorted(glob.glob("/opt/rocm/lib/librocsparse.so.*"), key=os.path.getsize)[-1]
        mib = int(name[9:])
        return np.fromfile(p, dtype=np.uint8, count=mib << 20, offset=(64 << 20) if mib > 64 else 0)
    raise SystemExit("unknown case " + name)


L = _abi.load(); s = HipSuffixSort(0)
case = sys.argv[1]
T = np.ascontiguousarray(load_case(case)); n = T.size
dT = torch.from_numpy(T).cuda()
first = None
for v in (sys.argv[2:] or [""]):
    sets = dict(kv.split("=") for kv in v.split(",") if kv)
    for k, val in sets.items(): os.environ[k] = val
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    s.Sort(dT, out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); s.Sort(dT, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    L.dq_profile_enable(1); L.dq_profile_reset()
    s.Sort(dT, out); torch.cuda.synchronize()
    L.dq_profile_enable(0)
    print(f"== {case} [{v or 'defaults'}] {min(ts)*1e3:.2f} ms (min of 3; {', '.join(f'{x*1e3:.2f}' for x in ts)})  {_abi.last_sort_info()}", flush=True)
    for k, p in _abi.profile_snapshot().items():
        if p["launches"]:
            print(f"   {k:26s} launches={p['launches']:4d} total={p['ms']:8.3f} ms  alg={p['alg_bytes']/max(p['ms'],1e-9)/1e6:8.1f} GB/s", flush=True)
    if os.environ.get("T_TRACE"):
        os.environ["DQ_TRACE"] = "1"; s.Sort(dT, out); torch.cuda.synchronize(); del os.environ["DQ_TRACE"]
    if first is None:
        first = out
        SA = out.cpu().numpy()
        print("   sufcheck", oracle.sufcheck_mt(T, SA), "sampled", oracle.verify_sampled(T, SA, 1_000_000, 3), flush=True)
    else:
        print("   equals the first variant's SA:", bool(torch.equal(first, out)), flush=True)
    for k in sets: del os.environ[k]
