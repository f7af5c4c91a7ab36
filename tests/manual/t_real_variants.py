#!/usr/bin/env python3
"""Real data from the image under forced settings: device-resident time, rounds; every variant's SA equals the first's.
usage: t_real_variants.py "" "DQ_RUNS=0" ..."""
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
        return np.frombuffer(f.read(nbytes), dtype=np.uint8).copy()

s = HipSuffixSort(0)
libs = sorted(glob.glob("/opt/rocm/lib/librocsparse.so.*"), key=os.path.getsize)
torchlib = sorted(glob.glob("/usr/local/lib/python3*/dist-packages/torch/lib/libtorch_cpu.so"))
cases = []
if libs:
    cases.append(("librocsparse.so, 64 MiB from offset 0", read_prefix(libs[-1], 64 << 20)))
    cases.append(("librocsparse.so, 256 MiB from offset 64 MiB", read_prefix(libs[-1], 256 << 20, 64 << 20)))
if torchlib:
    cases.append(("libtorch_cpu.so, first 128 MiB", read_prefix(torchlib[0], 128 << 20)))
# a sparse / padded image: blocks of data between runs of padding of every length (not periodic)
rng = np.random.default_rng(12)
parts, size = [], 0
while size < (64 << 20):
    parts.append(rng.integers(0, 256, int(rng.integers(1, 65536)), dtype=np.uint8))
    parts.append(np.full(int(rng.integers(1, 262144)), int(rng.choice([0, 0, 0, 255])), np.uint8))
    size += parts[-1].size + parts[-2].size
cases.append(("sparse image: random blocks <= 64 KiB between runs <= 256 KiB of 0x00 / 0xFF, 64 MiB", np.concatenate(parts)[:64 << 20].copy()))
text = None
try:
    from tools import datagen
    text = datagen.gen_enwik_like(48 << 20, 5, 65536)
    parts, pos = [], 0
    while pos < text.size:
        ln = int(rng.integers(1000, 200_000))
        parts.append(text[pos:pos + ln]); pos += ln
        parts.append(np.zeros(int(rng.integers(1, 100_000)), np.uint8))
    cases.append(("tar-like: text members of 1 - 200 kB, each followed by zero padding <= 100 kB", np.concatenate(parts)[:64 << 20].copy()))
except Exception as e:
    print("no datagen:", e)
variants = sys.argv[1:] or [""]
for name, T in cases:
    n = T.size
    dT = torch.from_numpy(T).cuda(); out = torch.empty(n, dtype=torch.int32, device="cuda")
    first = None
    for v in variants:
        sets = dict(kv.split("=") for kv in v.split(",") if kv)
        for k, val in sets.items(): os.environ[k] = val
        s.Sort(dT, out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); s.Sort(dT, out); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        same = True
        if first is None:
            first = out.clone()
            SA = out.cpu().numpy()
            chk = f"sufcheck={oracle.sufcheck_mt(T, SA)} sampled={oracle.verify_sampled(T, SA, 200_000, 3)}"
        else:
            same = bool(torch.equal(first, out)); chk = ""
        print(f"{name}: n={n} [{v or 'defaults'}] {min(ts)*1e3:.2f} ms {_abi.last_sort_info()} same={same} {chk}", flush=True)
        for k in sets: del os.environ[k]
