#!/usr/bin/env python3
"""bench.py -- MB of text suffix-sorted per second (bit-exact SA) on N MI355X.

One "step" = one ISuffixSort.Sort of the workload buffer, text already resident in HBM,
SA left in HBM (dq_sufsort_hip_dev_i32).  Workload at every N: BASELINE.json configs[1],
a 64 MiB uniform-random byte buffer per GPU (splitmix64, seed 0x5EED0002 + rank); the
path shards across independent inputs with no data-path collective, so N GPUs = N
buffers (weak scaling) and value = total MB sorted / max-over-ranks time.

    python bench.py [--gpus 1] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` is measured live: every radix_rank_kernel
launch in the timed region is bracketed by hipEvents on its launch stream (library-side,
dq_profile_*).  `cpu_baseline` times the oracle's single-threaded restatement of the
reference's LibDivSufSort on this host (rank 0, N=1 only) and bit-compares its SA with the
GPU's.  oracle/ is used here ONLY as that baseline/checker.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
SEED = 0x5EED0002


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size-mib", type=int, default=64, help="bytes of text per GPU (MiB)")
    ap.add_argument("--workload", choices=["uniform", "enwik"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel hipEvent timing")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import numpy as np
    import torch
    import torch.distributed as dist

    from deltaq_amd import HipSuffixSort, _abi, build as dq_build
    from deltaq_amd import workload as wl

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the suffix sorter has no CPU path")
    dq_build.build()                       # no-op when the in-tree .so is current
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    n = args.size_mib << 20
    seed = SEED + rank
    if args.workload == "uniform":
        host = wl.gen_uniform(n, seed)
        wname = f"{args.size_mib} MiB uniform-random bytes per GPU (splitmix64 seed 0x{SEED:X}+rank), int32 SA"
    else:
        from tools import datagen
        host = datagen.gen_enwik_like(n, 0xD17A0 + rank)
        wname = f"{args.size_mib} MiB enwik8-style skewed text per GPU (seed 0xD17A0+rank, R=256 KiB), int32 SA"
    text = torch.from_numpy(host).to(dev)
    sa = torch.empty(n, dtype=torch.int32, device=dev)
    sorter = HipSuffixSort(local_rank)
    L = _abi.load()

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        sorter.Sort(text, sa)
    L.dq_profile_reset()
    # timed region: hipEvents only around the dominant kernel (mode 2) -- bracketing all ~25
    # launches of a sort costs ~6 % of the step; the other kernels are profiled in an extra,
    # untimed pass below
    L.dq_profile_enable(0 if args.no_profile else 2)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sorter.Sort(text, sa)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    L.dq_profile_enable(0)
    prof = _abi.profile_snapshot()
    info = _abi.last_sort_info()
    prof_all, all_steps = {}, 3
    if not args.no_profile:
        L.dq_profile_reset()
        L.dq_profile_enable(1)
        for _ in range(all_steps):
            sorter.Sort(text, sa)
        torch.cuda.synchronize(dev)
        L.dq_profile_enable(0)
        prof_all = _abi.profile_snapshot()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        total_mb = world * n * args.steps / 1e6
        value = total_mb / elapsed
        rs = prof["radix_rank_kernel"]
        roofline = None
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.workload == "uniform" and args.size_mib == 64:
            # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
            # separate runs of this same command, tools/profile_gpu.sh): 2*FETCH_SIZE + WRITE_SIZE
            try:
                tj = json.load(open(tpath))
                traffic = int(tj["radix_rank_kernel"]["hbm_bytes_per_launch"])
                traffic_src = "profiles/traffic.json (rocprofv3 PMC, 2*FETCH_SIZE+WRITE_SIZE per launch)"
            except Exception:
                traffic = None
        if rs["launches"]:
            achieved = rs["alg_bytes"] / (rs["ms"] * 1e-3) / 1e9
            roofline = {
                "kernel": "radix_rank_kernel", "bound": "hbm",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "launches": rs["launches"],
                "avg_launch_us": round(rs["ms"] / rs["launches"] * 1e3, 2),
                "alg_bytes_per_launch": rs["alg_bytes"] // rs["launches"],
            }
        kernels = {k: {"launches_per_step": v["launches"] // all_steps, "ms_per_step": round(v["ms"] / all_steps, 4),
                       "alg_GBps": round(v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
                   for k, v in prof_all.items() if v["launches"]}
        out = {
            "metric": "MB of text suffix-sorted per second (bit-exact SA)",
            "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 text / u64 keys / int32 SA", "data": "synthetic",
            "config": {"workload": wname, "bytes_per_gpu": n, "residency": "text and SA resident in HBM",
                       "rounds_after_initial_sort": info["rounds"], "sharding": f"{world} independent buffers"},
            "roofline": roofline,
            "kernels_untimed_pass": kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, sa)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def cpu_baseline(host, sa_dev):
    """Single-threaded restatement of the reference's LibDivSufSort on this host, timed on the
    SAME buffer the GPU sorted (one run: ~8 s for 64 MiB), then bit-compared with the GPU SA."""
    import numpy as np
    import oracle
    n = host.size
    sample = host
    t0 = time.perf_counter()
    ref = oracle.divsufsort(sample)
    dt = time.perf_counter() - t0
    gpu = sa_dev.cpu().numpy()
    return {
        "value": round(n / 1e6 / dt, 3), "unit": "MB/s", "cores": 1, "kind": "port",
        "sample": f"the full {n >> 20} MiB workload buffer, 1 run, oracle/divsufsort.c (gcc -O2)",
        "seconds": round(dt, 3),
        "phases_s": {k: round(v, 3) for k, v in oracle.last_phase_seconds().items()},
        "host_cores_available": os.cpu_count(),
        "sa_bit_exact_vs_gpu": bool(np.array_equal(ref, gpu)),
    }


if __name__ == "__main__":
    sys.exit(main())
