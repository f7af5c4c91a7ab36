#!/usr/bin/env python3
"""bench.py -- MB of text suffix-sorted per second (bit-exact SA) on N MI355X.

One "step" = one ISuffixSort.Sort of the workload buffer with the text already resident in HBM
and the SA left in HBM (dq_sufsort_hip_dev_i32).  Workload: the north-star run of BASELINE.json,
a 256 MiB uniform-random byte buffer per GPU (splitmix64, seed 0x5EED0003 + rank), int32 SA.
A single suffix array does not shard; independent inputs do, with no
data-path collective, so N GPUs = N buffers (weak scaling) and
value = total MB sorted / max-over-ranks time.

    python bench.py [--gpus 1] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Beside the contract's fields it carries
  roofline      the dominant kernel, measured live: every launch of it inside the timed region is
                bracketed by hipEvents on its launch stream (library side, dq_profile_*)
  through_abi   the same buffer through the interface the reference exposes (host pointers in and
                out, dq_sufsort_hip_i32 == ISuffixSort.Sort(text, suffixes)): PCIe-inclusive, never `value`
  configs       one short record per other BASELINE.json configuration (64 MiB uniform, 256 MiB
                enwik-style text, and 128 x 16 MiB through the batch entry point on this rank's GPU)
  batch         (every N) BASELINE configs[4] as named: 128 x 16 MiB buffers (seeds 0x5EED0500 + j),
                LPT-sharded over the N ranks, host buffers in/out, MB/s = 2 GiB / max-over-ranks wall;
                for N > 1 the scatter/sort/gather layer also runs once over RCCL and rank 0 checks
                the gathered suffix arrays
  cpu_baseline  the oracle's single-threaded restatement of the reference's LibDivSufSort on this
                host (rank 0, N = 1 only), timed on the same 256 MiB buffer and bit-compared with the GPU's SA
oracle/ is used here ONLY as that baseline / checker.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6300.0          # ... and the copy ceiling measured on it (same guide; SURVEY.md section 8(d): report against both)
SEED_TARGET = 0x5EED0003       # 256 MiB uniform, the north-star run
SEED_CONFIG1 = 0x5EED0002      # 64 MiB uniform
SEED_ENWIK = 0xD17A0           # 256 MiB enwik-style text
SEED_BATCH = 0x5EED0500        # + j, 128 x 16 MiB
BATCH_COUNT, BATCH_BYTES = 128, 16 << 20


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size-mib", type=int, default=256, help="bytes of text per GPU (MiB)")
    ap.add_argument("--workload", choices=["uniform", "enwik"], default="uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel hipEvent timing")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region (profiling runs): no through_abi / configs / batch records")
    ap.add_argument("--no-build", action="store_true",
                    help="never spawn a compiler (runs under rocprofv3, where the GPU is up before main())")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL)")
    ap.add_argument("--batch-count", type=int, default=BATCH_COUNT, help="buffers of the batch record (configs[4]: 128)")
    ap.add_argument("--batch-mib", type=int, default=BATCH_BYTES >> 20, help="MiB per buffer of the batch record (configs[4]: 16)")
    ap.add_argument("--many-new-per-rank", type=int, default=4, help="new files per rank of the one-old-many-new record")
    ap.add_argument("--config3-cpu", action="store_true",
                    help="opt-in (minutes of one host core, not in the driver's default run): time the oracle's LibDivSufSort "
                         "restatement with 64-bit indices on the 2 GiB buffer of configs[3] and bit-compare (SURVEY 8(d) Config 4)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as fresh child
        # processes, before this process has loaded the library or touched the GPU; relay their output (rank 0
        # prints the line) and leave with their status.  One rank can never stand in for N.
        return launch_ranks(args.gpus, args.share_gpu)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree", file=sys.stderr)
        return 2

    # Native builds first, before anything initialises the GPU runtime (compilers are child processes;
    # under rocprofv3 the GPU is already up, so a stale library there is an error, not a rebuild).
    from deltaq_amd import build as dq_build
    import oracle
    from tools import datagen
    if args.no_build:
        if dq_build.is_stale():
            raise SystemExit("libdq_sufsort_hip.so is stale: run __graft_entry__.build() first")
    else:
        dq_build.build()                   # no-op when the in-tree .so is current
        oracle.build()
        datagen.build()

    managed = managed_reference_probe() if rank == 0 else None      # (child processes: before anything touches the GPU)

    import numpy as np
    import torch
    import torch.distributed as dist

    from deltaq_amd import HipSuffixSort, _abi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the suffix sorter has no CPU path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    devices_seen = [torch.cuda.current_device()]
    # one process per GPU: run on the CPUs of the NUMA node the device hangs off (host copies of the batch records
    # and the through-ABI timing stay on that socket); nothing happens where the platform does not say, or at N = 1
    # on a box whose only node it is anyway.  DQ_NUMA_BIND=0 turns it off.
    numa_node = _abi.bind_process_to_device_numa_node(local_rank) if world > 1 else _abi.load().dq_device_numa_node(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        # every rank names the device it runs on: one GPU per rank, unless the plumbing test shares cuda:0 on purpose
        seen = [None] * world
        dist.all_gather_object(seen, (int(torch.cuda.current_device()), os.environ.get("HIP_VISIBLE_DEVICES"),
                                      os.environ.get("ROCR_VISIBLE_DEVICES")))
        devices_seen = [x[0] for x in seen]
        nodes = [None] * world
        dist.all_gather_object(nodes, numa_node)
        numa_node = nodes
        if not args.share_gpu and len(set(seen)) != world:
            raise SystemExit(f"bench.py: {world} ranks but devices {seen}: two ranks share a GPU")

    n = args.size_mib << 20
    if args.workload == "uniform":
        seed = (SEED_TARGET if args.size_mib == 256 else SEED_CONFIG1) + rank
        host = datagen.gen_uniform(n, seed)
        wname = f"{args.size_mib} MiB uniform-random bytes per GPU (splitmix64 seed 0x{seed - rank:X}+rank), int32 SA"
    else:
        host = datagen.gen_enwik_like(n, SEED_ENWIK + rank)
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

    # which kernel dominates?  one profiled sort decides (its category is then the only one bracketed
    # by hipEvents inside the timed region: bracketing all ~25 launches of a sort costs ~6 % of the step)
    for _ in range(args.warmup):
        sorter.Sort(text, sa)
    dominant = "radix_rank_kernel"
    if not args.no_profile:
        L.dq_profile_reset()
        L.dq_profile_enable(1)
        sorter.Sort(text, sa)
        torch.cuda.synchronize(dev)
        L.dq_profile_enable(0)
        snap = _abi.profile_snapshot()
        dominant = max(snap, key=lambda k: snap[k]["ms"])
    L.dq_profile_reset()
    L.dq_profile_enable(0 if args.no_profile else 100 + _abi.category_of(dominant))
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

    extras = not args.no_extras
    through_abi = configs = None
    if extras and rank == 0:
        through_abi = time_through_abi(sorter, host, sa, reps=3)
    # the named configs[4] workload: every rank takes part
    batch = batch_config4(world, rank, local_rank, dev, args.backend, sorter, args.batch_count, args.batch_mib << 20, args.many_new_per_rank) if extras else None
    if extras and rank == 0 and world == 1:
        del text
        configs = other_configs(sorter, dev, args.config3_cpu)

    out = None
    if rank == 0:
        total_mb = world * n * args.steps / 1e6
        value = total_mb / elapsed
        rs = prof[dominant]
        roofline = None
        if rs["launches"]:
            achieved = rs["alg_bytes"] / (rs["ms"] * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(dominant, args)
            roofline = {
                "kernel": dominant, "bound": "hbm",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_of_6p3": round(achieved / HBM_COPY_GBS, 4),      # against the measured copy ceiling
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_stale": bool(traffic is None and traffic_src is not None and traffic_src.startswith("stale")),
                "launches": rs["launches"],
                "avg_launch_us": round(rs["ms"] / rs["launches"] * 1e3, 2),
                "alg_bytes_per_launch": rs["alg_bytes"] // rs["launches"],
            }
        kernels = kernel_table(prof_all, all_steps)
        out = {
            "metric": "MB of text suffix-sorted per second (bit-exact SA)",
            "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "gpus_flag": args.gpus, "devices_per_rank": devices_seen,
            "numa_node_per_rank": numa_node if isinstance(numa_node, list) else [numa_node],
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 text / u64 keys / int32 SA", "data": "synthetic",
            "config": {"workload": wname, "bytes_per_gpu": n, "residency": "text and SA resident in HBM",
                       "rounds_after_initial_sort": info["rounds"], "sharding": f"{world} independent buffers"},
            "roofline": roofline,
            "kernels_untimed_pass": kernels,
        }
        if through_abi is not None:
            out["through_abi"] = through_abi
        if configs is not None:
            out["configs"] = configs
        if batch is not None:
            out["batch"] = batch
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(host, sa)
            cb["managed_reference"] = managed
            out["cpu_baseline"] = cb
            if through_abi is not None:
                through_abi["speedup_vs_cpu"] = round(through_abi["MBps"] / cb["value"], 1)
            out["speedup_vs_cpu_device_resident"] = round(value / cb["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def visible_gpu_count():
    """GPUs this node shows, counted WITHOUT loading torch or HIP in this process (the launcher must not open the
    device: torch.cuda.device_count() falls through to hipGetDeviceCount on builds without amdsmi -- round-5 advice):
    the KFD topology's nodes with SIMDs, cut down by HIP_/ROCR_VISIBLE_DEVICES when set.  None = cannot tell."""
    import glob
    count = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for p in nodes:
        try:
            with open(p) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            count += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            count = min(count, len([x for x in v.split(",") if x.strip() != ""]))
    return count


def launch_ranks(n, share_gpu=False):
    """`python bench.py --gpus N` with no launcher around it: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a child process (nothing in
    this process has initialised the GPU -- or even imported torch), pass its output through and return its exit status."""
    import socket
    import subprocess
    if not os.environ.get("DQ_BENCH_ALLOW_OVERSUBSCRIBE") and not share_gpu:
        have = visible_gpu_count()
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} but this node shows {have} GPU(s)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print("bench.py: no launcher around --gpus %d, starting the ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, cwd=ROOT)


def managed_reference_probe():
    """"if dotnet happens to exist on the GPU box, additionally time the real managed
    LibDivSufSort".  The probe runs before the GPU runtime is initialised (it starts a child process).  The managed
    provider's sources are the reference's (never copied into this repository, and /root/reference does not exist on
    the GPU box), so a toolchain alone is not enough: the harness also needs DQ_REFERENCE_DIR to point at a checkout."""
    import shutil
    import subprocess
    exe = shutil.which("dotnet") or shutil.which("mono")
    rec = {"toolchain": exe, "timed": False}
    if not exe:
        rec["note"] = "no dotnet / mono on this host: the C restatement (kind 'port') is the CPU baseline"
        return rec
    try:
        rec["version"] = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=60).stdout.strip()[:80]
    except Exception as e:              # noqa: BLE001
        rec["version"] = repr(e)[:80]
    ref = os.environ.get("DQ_REFERENCE_DIR")
    proj = os.path.join(ref, "src", "DeltaQ.SuffixSorting.LibDivSufSort") if ref else None
    if not (proj and os.path.isdir(proj)):
        rec["note"] = "toolchain present, reference checkout absent (set DQ_REFERENCE_DIR): managed LibDivSufSort not timed"
        return rec
    rec["note"] = ("toolchain and reference checkout present: build bindings/csharp/DeltaQ.SuffixSorting.Hip.Tests against it and "
                   "run its benchmark harness (INTEGRATION.md section 2); not automated in bench.py")
    return rec


def kernel_table(snapshot, steps):
    """Per kernel category: launches and time per step, algorithmic GB/s (DESIGN.md section 4's bytes per element x
    the elements of each launch, summed by the library) and its fraction of the 8 TB/s HBM peak."""
    return {k: {"launches_per_step": round(v["launches"] / steps, 2), "ms_per_step": round(v["ms"] / steps, 4),
                "alg_GBps": round(v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6, 1),
                "frac_of_hbm_peak": round(v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
                "frac_of_6p3": round(v["alg_bytes"] / max(v["ms"], 1e-9) / 1e6 / HBM_COPY_GBS, 4)}
            for k, v in snapshot.items() if v["launches"]}


def profiled_kernels(sorter, host, dev, steps=2):
    """kernel_table() of `steps` sorts of `host` with every launch bracketed by hipEvents (a pass of its own: the
    events cost ~6 % of a step, so this is never the timed region)."""
    import torch
    from deltaq_amd import _abi
    L = _abi.load()
    text = torch.from_numpy(host).to(dev)
    sa = torch.empty(host.size, dtype=torch.int32, device=dev)
    sorter.Sort(text, sa)
    L.dq_profile_reset()
    L.dq_profile_enable(1)
    for _ in range(steps):
        sorter.Sort(text, sa)
    torch.cuda.synchronize(dev)
    L.dq_profile_enable(0)
    return kernel_table(_abi.profile_snapshot(), steps)


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, tools/profile_gpu.sh):
    2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md.
    The file names the library it was taken from (deltaq_amd.build._source_digest() at profiling time): counters of
    another build are not reported -- (None, "stale: ...") instead."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    try:
        from deltaq_amd import build as dq_build
        tj = json.load(open(tpath))
        if tj.get("workload") != f"{args.workload}-{args.size_mib}MiB":
            return None, None
        have, want = tj.get("library_source_digest"), dq_build._source_digest()
        if have != want:
            return None, f"stale: profiles/traffic.json was taken from library sources {str(have)[:12]}, this is {want[:12]}"
        return int(tj[kernel]["hbm_bytes_per_launch"]), \
            "profiles/traffic.json (rocprofv3 PMC, 2*FETCH_SIZE+WRITE_SIZE per launch; library sources %s)" % want[:12]
    except Exception:
        return None, None


def time_device(sorter, host, dev, reps):
    import torch
    text = torch.from_numpy(host).to(dev)
    sa = torch.empty(host.size, dtype=torch.int32, device=dev)
    sorter.Sort(text, sa)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        sorter.Sort(text, sa)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps * 1e3


def time_through_abi(sorter, host, sa_dev=None, reps=3):
    """ISuffixSort.Sort(text, suffixes) as the reference's caller sees it: pageable host memory in and out."""
    import numpy as np
    n = host.size
    out = np.ones(n, dtype=np.int32)            # touched pages, like a pooled MemoryOwner<int>
    sorter.Sort(host, out)                      # first call grows the workspace
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sorter.Sort(host, out)
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    rec = {"entry": "dq_sufsort_hip_i32 (host pointers in/out, PCIe-inclusive)", "ms": round(best * 1e3, 3),
           "MBps": round(n / 1e6 / best, 1), "reps": reps}
    if sa_dev is not None:
        rec["sa_equals_device_resident"] = bool(np.array_equal(out, sa_dev.cpu().numpy()))
    return rec


def other_configs(sorter, dev, config3_cpu=False):
    """One short record per BASELINE.json configuration that is not the timed workload."""
    from deltaq_amd import _abi
    from tools import datagen
    recs = []
    for name, gen, reps in (
            ("configs[1]: 64 MiB uniform random", lambda: datagen.gen_uniform(64 << 20, SEED_CONFIG1), 10),
            ("configs[2]: 256 MiB enwik8-style text", lambda: datagen.gen_enwik_like(256 << 20, SEED_ENWIK), 3)):
        host = gen()
        ms = time_device(sorter, host, dev, reps)
        rounds = _abi.last_sort_info()["rounds"]
        abi = time_through_abi(sorter, host, None, reps=2)
        recs.append({"config": name, "device_resident_ms": round(ms, 3), "device_resident_MBps": round(host.size / 1e3 / ms, 1),
                     "through_abi_ms": abi["ms"], "through_abi_MBps": abi["MBps"], "rounds": rounds})
        if "enwik" in name:
            recs[-1]["kernels"] = profiled_kernels(sorter, host, dev)
        del host
    recs.append(real_binary_record(sorter, dev))
    recs.append(match_search_record(sorter, dev))
    recs.append(reference_benchmark_shape(sorter))
    recs.append(bsdiff_create_record(dev))
    recs.append(config3_record(sorter, dev, cpu=config3_cpu))         # last: it leaves > 100 GiB of cached workspace, released at its end
    return recs


SEED_CONFIG3 = 0x5EED0004      # 2 GiB uniform, int64 SA


def config3_record(sorter, dev, reps=3, cpu=False):
    """BASELINE configs[3]: 2 GiB uniform-random bytes, 64-bit suffix array (dq_sufsort_hip_dev_i64), text and SA
    resident in HBM.  The suffix array of the timed sorts is checked on the host by LDSSChecker.Check (the oracle's
    threaded evaluation) and 10^5 sampled strict pairs.  Skipped, with the reason, where the device or the host is
    too small for it (text + 16 GiB of SA + ~112 GiB of workspace in HBM; text + SA + the checker's arrays in RAM)."""
    import numpy as np
    import torch
    import oracle
    from deltaq_amd import _abi
    from tools import datagen
    name = "configs[3]: 2 GiB uniform random, int64 SA"
    n = 1 << 31
    L = _abi.load()
    try:
        free_hbm, _total = torch.cuda.mem_get_info(dev)
        need_hbm = n + 8 * n + int(L.dq_sufsort_hip_workspace_bytes(n, 8)) + (1 << 30)
        ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_AVPHYS_PAGES")
        need_ram = n + 8 * n + 8 * n + (4 << 30)           # text, SA, the checker's inverse, slack
        L.dq_sufsort_hip_release()                         # the earlier records' cached workspaces go first
        torch.cuda.empty_cache()
        free_hbm, _total = torch.cuda.mem_get_info(dev)
        if free_hbm < need_hbm:
            return {"config": name, "skipped": f"free HBM {free_hbm >> 30} GiB < {need_hbm >> 30} GiB needed"}
        if ram < need_ram:
            return {"config": name, "skipped": f"free host RAM {ram >> 30} GiB < {need_ram >> 30} GiB needed for the check"}
        host = datagen.gen_uniform(n, SEED_CONFIG3)
        text = torch.from_numpy(host).to(dev)
        sa = torch.empty(n, dtype=torch.int64, device=dev)
        sorter.Sort(text, sa)                              # first call grows the workspace
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            sorter.Sort(text, sa)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        info = _abi.last_sort_info()
        ms = min(ts) * 1e3
        rec = {"config": name, "device_resident_ms": round(ms, 3), "device_resident_ms_all": [round(x * 1e3, 2) for x in ts],
               "device_resident_MBps": round(n / 1e3 / ms, 1), "rounds": info["rounds"], "index_bytes": 8}
        L.dq_profile_reset()
        L.dq_profile_enable(1)
        sorter.Sort(text, sa)
        torch.cuda.synchronize(dev)
        L.dq_profile_enable(0)
        rec["kernels"] = kernel_table(_abi.profile_snapshot(), 1)
        got = np.empty(n, dtype=np.int64)
        step = 1 << 28
        for a in range(0, n, step):                        # in pieces: no pinned 16 GiB staging buffer
            got[a:a + step] = sa[a:a + step].cpu().numpy()
        del sa, text
        t0 = time.perf_counter()
        rec["sufcheck_mt"] = int(oracle.sufcheck_mt(host, got))
        rec["sampled_strict_pairs_1e5_first_bad"] = int(oracle.verify_sampled(host, got, 100_000, 11))
        rec["check_s"] = round(time.perf_counter() - t0, 1)
        rec["checked_ok"] = bool(rec["sufcheck_mt"] == oracle.CHECK_DONE and rec["sampled_strict_pairs_1e5_first_bad"] == -1)
        if cpu:
            # SURVEY.md section 8(d), Config 4: "CPU baseline = restatement instantiated with 64-bit indices (the reference
            # itself cannot represent this input)" -- ISuffixSort.cs:18,27 are `int`.  One host core, once.
            t0 = time.perf_counter()
            ref = oracle.divsufsort(host, dtype=np.int64)
            cs = time.perf_counter() - t0
            rec["cpu_restatement_i64"] = {"seconds": round(cs, 1), "MBps": round(n / 1e6 / cs, 2), "cores": 1, "kind": "port",
                                          "bit_exact_vs_gpu": bool(np.array_equal(ref, got)),
                                          "gpu_speedup_device_resident": round(cs * 1e3 / ms, 0)}
            del ref
        return rec
    except Exception as e:                                  # noqa: BLE001 -- report, do not lose the bench line
        return {"config": name, "error": repr(e)[:300]}
    finally:
        L.dq_sufsort_hip_release()
        torch.cuda.empty_cache()


def real_binary_record(sorter, dev, mib=128):
    """What `dq bsdiff` is pointed at (the reference's README diffs executables): the first 128 MiB of a real shared
    library of this image -- machine code, tables, strings, code built for several targets -- not generator output.
    Device-resident time with its per-kernel table, and the suffix array bit-compared with the oracle's LibDivSufSort."""
    import glob
    import numpy as np
    import oracle
    from deltaq_amd import _abi
    cands = sorted(glob.glob("/usr/local/lib/python3*/dist-packages/torch/lib/libtorch_cpu.so")) + \
        sorted(glob.glob("/opt/rocm/lib/librocsparse.so.*"), key=os.path.getsize, reverse=True)
    path = next((p for p in cands if os.path.getsize(p) >= (mib << 20)), None)
    if path is None:
        return {"config": "real binary: a shared library of the image", "skipped": "no library of >= %d MiB found" % mib}
    host = np.fromfile(path, dtype=np.uint8, count=mib << 20)
    ms = time_device(sorter, host, dev, 3)
    info = _abi.last_sort_info()
    rec = {"config": f"real binary: first {mib} MiB of {os.path.basename(path)} (a shared library of this image)",
           "device_resident_ms": round(ms, 3), "device_resident_MBps": round(host.size / 1e3 / ms, 1),
           "rounds": info["rounds"], "sum_active_over_n": round(info["sum_active"] / host.size, 2),
           "kernels": profiled_kernels(sorter, host, dev)}
    import torch
    d = torch.from_numpy(host).to(dev)
    got = sorter.Sort(d).cpu().numpy()
    t0 = time.perf_counter()
    ref = oracle.divsufsort(host)
    ct = time.perf_counter() - t0
    rec["cpu_oracle_s"] = round(ct, 2)
    rec["bit_exact_vs_oracle"] = bool(np.array_equal(got, ref))
    return rec


def bsdiff_create_record(dev):
    """Diff.Create natively (dq_bsdiff_create: device suffix array, device match search
    in windows under the reference's scan loop, bzip2 framing with the block transform on the device sorter) on a
    16 MiB random file and a copy with 2000 small edits, beside the CPU pipeline it replaces (oracle restatements
    of LibDivSufSort and of the Search / scan loop on one core, libbz2)."""
    import bz2
    import numpy as np
    import oracle
    from deltaq_amd import Diff, Patch
    from tools import datagen
    rng = np.random.default_rng(3)
    old = datagen.gen_uniform(16 << 20, 5)
    new = bytearray(old.tobytes())
    for _ in range(2000):
        k, a, ln = int(rng.integers(0, 3)), int(rng.integers(0, len(new))), int(rng.integers(1, 400))
        if k == 0:
            new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1:
            del new[a:a + ln]
        else:
            new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
    new = np.frombuffer(bytes(new), dtype=np.uint8)
    Diff.CreateBytes(old[:4096], new[:4096], dev.index)
    t0 = time.perf_counter()
    patch = Diff.CreateBytes(old, new, dev.index)
    gt = time.perf_counter() - t0
    ctrl, diff, extra, stats = Diff.Scan(old, new, dev.index)
    t0 = time.perf_counter()
    sa = oracle.divsufsort(old)
    t1 = time.perf_counter()
    wc, wd, we, _ = oracle.bsdiff_scan(old, sa, new)
    t2 = time.perf_counter()
    for x in (wc, wd, we):
        bz2.compress(x.tobytes())
    t3 = time.perf_counter()
    return {"config": "Diff.Create natively (BSDIFF40): 16 MiB random old, new = old with 2000 small edits",
            "create_ms": round(gt * 1e3, 1), "patch_bytes": len(patch), "scan_loop": stats,
            "cpu_pipeline_ms": {"sort": round((t1 - t0) * 1e3), "scan": round((t2 - t1) * 1e3), "bzip2": round((t3 - t2) * 1e3)},
            "raw_streams_equal_oracle": bool(np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we)),
            "patch_applies": bool(Patch.Apply(old, patch) == new.tobytes())}


def reference_benchmark_shape(sorter):
    """The reference's own benchmark (bench/DeltaQ.Benchmarks/SuffixSortingBenchmarks.cs:27-65): Sort(asset) on
    new Random(670761).NextBytes(size) buffers of 0 ... 32 KiB and 64 KiB ... 1 MiB; here a few of those sizes,
    through the host interface (what BenchmarkDotNet would time), median of 20 calls, beside the oracle's
    LibDivSufSort restatement on one host core."""
    import numpy as np
    import oracle
    rows = []
    for size in (4096, 32768, 65536, 262144, 1048576):
        T = oracle.net_random_bytes(size)
        sa = np.ones(size, np.int32)
        sorter.Sort(T, sa)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            sorter.Sort(T, sa)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ref = oracle.divsufsort(T)
        ct = time.perf_counter() - t0
        rows.append({"bytes": size, "through_abi_us": round(sorted(ts)[len(ts) // 2] * 1e6, 1),
                     "cpu_oracle_us": round(ct * 1e6, 1), "bit_exact": bool(np.array_equal(ref, sa))})
    # the same sizes with text-like content (the reference's users diff executables and text, not noise): device-wide
    # pipeline with its doubling rounds instead of one radix sort
    from tools import datagen
    text_rows = []
    for size in (65536, 262144, 1048576, 4194304):
        T = datagen.gen_enwik_like(size, SEED_ENWIK, 65536)
        sa = np.ones(size, np.int32)
        sorter.Sort(T, sa)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            sorter.Sort(T, sa)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        ref = oracle.divsufsort(T)
        ct = time.perf_counter() - t0
        text_rows.append({"bytes": size, "through_abi_us": round(sorted(ts)[len(ts) // 2] * 1e6, 1),
                          "cpu_oracle_us": round(ct * 1e6, 1), "bit_exact": bool(np.array_equal(ref, sa))})
    return {"config": "reference benchmark shape: Sort(new Random(670761).NextBytes(size)), host interface", "sizes": rows,
            "text_like_sizes": text_rows}


def match_search_record(sorter, dev):
    """SURVEY.md section 8(f) row 1, the sort's consumer: Diff.cs Search for 10^6 consecutive scan positions of a
    differing region (two independent 16 MiB random buffers), suffix array left on the device by the sorter;
    the oracle's restatement of Search answers a sample of them on one host core."""
    import numpy as np
    import torch
    import oracle
    from deltaq_amd import HipMatchSearch, _abi
    from tools import datagen
    L = _abi.load()
    n, count, sample = 16 << 20, 1_000_000, 100_000
    old, new = datagen.gen_uniform(n, SEED_BATCH), datagen.gen_uniform(n, SEED_BATCH + 1)
    d_old, d_new = torch.from_numpy(old).to(dev), torch.from_numpy(new).to(dev)
    d_sa = sorter.Sort(d_old)
    ms = HipMatchSearch(dev.index)
    ms.Search(d_sa, d_old, d_new, scan0=0, count=1000)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    pos, ln = ms.Search(d_sa, d_old, d_new, scan0=0, count=count)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    sa = d_sa.cpu().numpy()
    t0 = time.perf_counter()
    wpos, wlen = oracle.bsdiff_search(old, sa, new, scan0=0, count=sample)
    ct = time.perf_counter() - t0
    ok = bool(np.array_equal(pos[:sample].cpu().numpy(), wpos) and np.array_equal(ln[:sample].cpu().numpy(), wlen))
    return {"config": "match search (Diff.cs:267-298 Search) on the device-resident SA: 16 MiB old, 10^6 scan positions",
            "device_resident_ms": round(dt * 1e3, 3), "M_queries_per_s": round(count / dt / 1e6, 1),
            "cpu_oracle_M_queries_per_s": round(sample / ct / 1e6, 3), "cores": 1,
            "bit_exact_vs_oracle_sample": ok}


def batch_config4(world, rank, local_rank, dev, backend, sorter, BATCH_COUNT=BATCH_COUNT, BATCH_BYTES=BATCH_BYTES, many_per_rank=4):
    """BASELINE configs[4]: 128 x 16 MiB independent buffers, LPT-sharded over the ranks, host buffers in
    and out through dq_sufsort_hip_batch_i32 on each rank's GPU; no data-path collective.  Timed twice: with the
    pageable buffers a P/Invoke caller hands over, and with page-locked ones (what separates PCIe / host-memory
    contention between the ranks of a node from GPU time: per-rank host-copy GB/s are listed).  Then, for N > 1, the
    two exchange layers once each over the process group (RCCL under nccl): scatter / sort / gather of a small
    batch, and one old file + many new files (broadcast of text and suffix array, deltaq_amd.batch)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from deltaq_amd import _abi
    from deltaq_amd.batch import plan_shards, sort_batch_distributed
    from tools import datagen
    L = _abi.load()
    plan = plan_shards([BATCH_BYTES] * BATCH_COUNT, world)
    mine = plan[rank]
    cnt = len(mine)
    texts = [datagen.gen_uniform(BATCH_BYTES, SEED_BATCH + j) for j in mine]
    sas = [np.ones(BATCH_BYTES, np.int32) for _ in mine]           # pre-touched output pages
    devs = (ctypes.c_int32 * 1)(local_rank)
    ln = (ctypes.c_int64 * cnt)(*[BATCH_BYTES] * cnt)
    cdev = dev if backend == "nccl" else "cpu"

    def run(tx, sx):
        tp = (ctypes.c_void_p * cnt)(*[t.ctypes.data for t in tx])
        sp = (ctypes.c_void_p * cnt)(*[a.ctypes.data for a in sx])
        # warm the pipeline's slots and the workspace
        _abi.check(L.dq_sufsort_hip_batch_i32(min(cnt, 3), tp, ln, sp, 1, devs))
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        _abi.check(L.dq_sufsort_hip_batch_i32(cnt, tp, ln, sp, 1, devs))
        mine_s = time.perf_counter() - t0
        bi = _abi.last_batch_info()                  # busy time of this rank's three pipeline stages
        row = [mine_s, bi["copy_in_ms"], bi["sort_ms"], bi["copy_out_ms"], float(bi["shares_bound_to_numa_node"])]
        rows = [row]
        if world > 1:
            t = torch.zeros(world, len(row), dtype=torch.float64, device=cdev)
            t[rank] = torch.tensor(row, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            rows = [[float(x) for x in r] for r in t.tolist()]
        stage_rows.append(rows)
        return [r[0] for r in rows]

    stage_rows = []
    walls = run(texts, sas)
    wall = max(walls)
    # the same call on page-locked buffers (torch's pinned allocator; the library sees plain pointers)
    pin_walls = None
    try:
        ptexts = [torch.from_numpy(t).pin_memory() for t in texts]
        psas = [torch.empty(BATCH_BYTES, dtype=torch.int32).pin_memory() for _ in mine]
        pin_walls = run([t.numpy() for t in ptexts], [a.numpy() for a in psas])
        pinned_ok = all(np.array_equal(a.numpy(), b) for a, b in list(zip(psas, sas))[:2])
        del ptexts, psas
    except Exception as e:                              # pinned memory exhausted on a small host: report, go on
        pinned_ok = repr(e)[:200]
    # device-resident rate of the same buffers (what `value` would be for this shape)
    dms = time_device(sorter, texts[0], dev, 8) if cnt else 0.0
    rec = None
    if rank == 0:
        import oracle
        ok = all(oracle.sufcheck_mt(texts[k], sas[k]) == 0 for k in range(min(cnt, 4)))
        per_rank_bytes = [len(p) * BATCH_BYTES * 5 for p in plan]                      # text in + SA out
        named = BATCH_COUNT == 128 and BATCH_BYTES == 16 << 20
        rec = {"workload": ("BASELINE configs[4]: " if named else "NOT configs[4] (a plumbing run): ") +
                           f"{BATCH_COUNT} x {BATCH_BYTES >> 20} MiB uniform random (seeds 0x5EED0500+j), int32 SA",
               "entry": "dq_sufsort_hip_batch_i32 per rank (host pointers in/out, PCIe-inclusive)",
               "sharding": {"policy": "LPT over ranks (deltaq_amd.batch.plan_shards)",
                            "buffers_per_rank": [len(p) for p in plan]},
               "wall_ms": round(wall * 1e3, 2), "MBps": round(BATCH_COUNT * BATCH_BYTES / 1e6 / wall, 1),
               "host_copy_GBps_per_rank": [round(b / w / 1e9, 2) for b, w in zip(per_rank_bytes, walls)],
               # busy time of each rank's pipeline stages (dq_last_batch_info): a rank whose sort_ms is close to its wall
               # waits for its GPU, one whose copy stages are waits for host memory / PCIe
               "wall_ms_per_rank": [round(r[0] * 1e3, 2) for r in stage_rows[0]],
               "copy_in_ms_per_rank": [round(r[1], 2) for r in stage_rows[0]],
               "sort_ms_per_rank": [round(r[2], 2) for r in stage_rows[0]],
               "copy_out_ms_per_rank": [round(r[3], 2) for r in stage_rows[0]],
               "ranks_with_numa_bound_threads": int(sum(1 for r in stage_rows[0] if r[4] > 0)),
               "device_resident_ms_per_buffer": round(dms, 3),
               "device_resident_MBps_per_gpu": round(BATCH_BYTES / 1e3 / dms, 1) if dms else None,
               "sufcheck_first_buffers": bool(ok)}
        if pin_walls is not None:
            rec["pinned_buffers"] = {"wall_ms": round(max(pin_walls) * 1e3, 2),
                                     "MBps": round(BATCH_COUNT * BATCH_BYTES / 1e6 / max(pin_walls), 1),
                                     "host_copy_GBps_per_rank": [round(b / w / 1e9, 2) for b, w in zip(per_rank_bytes, pin_walls)],
                                     "sort_ms_per_rank": [round(r[2], 2) for r in stage_rows[-1]],
                                     "same_suffix_arrays": pinned_ok}
        if world == 1:
            rec["cpu_replicas"] = cpu_replica_baseline(texts)
    del texts, sas
    if world > 1:
        # the gather layer over RCCL (torch.distributed "nccl"): 2 small buffers per rank, SAs to rank 0
        small = None
        if rank == 0:
            small = [datagen.gen_uniform((1 << 20) + 4099 * j, SEED_BATCH + 1000 + j) for j in range(2 * world)]
        try:
            got = sort_batch_distributed(small, gather_to_root=True)
            if rank == 0:
                import oracle
                good = all(oracle.sufcheck_mt(t, np.ascontiguousarray(s)) == 0 for t, s in zip(small, got))
                rec["rccl_scatter_sort_gather"] = {"backend": backend, "buffers": len(small), "sufcheck": bool(good)}
        except Exception as e:                      # report, do not lose the bench line
            if rank == 0:
                rec["rccl_scatter_sort_gather"] = {"backend": backend, "error": repr(e)[:300]}
    many = one_old_many_new(world, rank, dev, backend, many_per_rank)
    if rank == 0 and many is not None:
        rec["one_old_many_new"] = many
    return rec


def one_old_many_new(world, rank, dev, backend, per_rank=4):
    """The many-files bsdiff path with its one exchange step: a 16 MiB old file, 4 new files per rank (copies with
    ~200 small edits).  Rank 0 sorts the old file once, text + suffix array are broadcast (RCCL under nccl), every
    rank diffs its share against a DiffIndex on the received buffers, patches are gathered and applied on rank 0.
    Beside it: the same diffs with the old file sorted again for every pair (Diff.Create as the reference calls it)."""
    import numpy as np
    import torch.distributed as dist
    from deltaq_amd import Diff, DiffIndex, Patch
    from deltaq_amd.batch import diff_many_distributed
    from tools import datagen
    old = news = None
    if rank == 0:
        rng = np.random.default_rng(9)
        old = datagen.gen_uniform(16 << 20, SEED_BATCH + 77)
        news = []
        for j in range(per_rank * world):
            x = bytearray(old.tobytes())
            for _ in range(200):
                k, a, ln = int(rng.integers(0, 3)), int(rng.integers(0, len(x))), int(rng.integers(1, 300))
                if k == 0:
                    x[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
                elif k == 1:
                    del x[a:a + ln]
                else:
                    x[a:a + ln] = rng.integers(0, 256, min(ln, len(x) - a), dtype=np.uint8).tobytes()
            news.append(np.frombuffer(bytes(x), dtype=np.uint8))
    try:
        if world > 1:
            dist.barrier()
            t0 = time.perf_counter()
            patches = diff_many_distributed(old, news)
            dt = time.perf_counter() - t0
        else:
            Diff.CreateBytes(old, news[0], dev.index)      # (both forms are timed warm: buffers grown, pinned areas there)
            dt = None
            for _ in range(2):                             # (the second pass: emitter threads' buffers and the index's exist)
                t0 = time.perf_counter()
                with DiffIndex(old, dev.index) as ix:
                    patches = [ix.Create(x) for x in news]
                dt = time.perf_counter() - t0
        if rank != 0:
            return None
        t0 = time.perf_counter()
        again = [Diff.CreateBytes(old, x, dev.index) for x in news[:per_rank]]
        per_pair = (time.perf_counter() - t0) / per_rank
        ok = all(Patch.Apply(old, p) == x.tobytes() for p, x in zip(patches, news))
        return {"old_bytes": int(old.size), "new_files": len(news), "exchange": "broadcast of text + suffix array, gather of patches"
                if world > 1 else "none (one rank): DiffIndex built once",
                "backend": backend if world > 1 else None, "wall_ms": round(dt * 1e3, 1),
                "ms_per_new_file": round(dt * 1e3 / len(news), 1),
                "ms_per_pair_with_its_own_sort": round(per_pair * 1e3, 1),
                "patches_equal_per_pair_create": bool(patches[:per_rank] == again), "patches_apply": bool(ok)}
    except Exception as e:                              # report, do not lose the bench line
        return {"error": repr(e)[:300]} if rank == 0 else None


def cpu_replica_baseline(texts):
    """the generous CPU figure for the batch shape -- one 16 MiB buffer per host core at the
    same time (the oracle's LibDivSufSort restatement releases the GIL inside its C call), as many replicas as cores,
    at most the 128 buffers of the config."""
    import concurrent.futures as cf
    import oracle
    k = max(1, min(os.cpu_count() or 1, len(texts), BATCH_COUNT))
    oracle.divsufsort(texts[0][:1 << 16])
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=k) as ex:
        list(ex.map(oracle.divsufsort, texts[:k]))
    dt = time.perf_counter() - t0
    return {"replicas": k, "cores": k, "seconds": round(dt, 3), "MBps": round(k * BATCH_BYTES / 1e6 / dt, 1),
            "kind": "port", "note": "oracle/divsufsort.c, one buffer per core concurrently"}


def cpu_baseline(host, sa_dev):
    """Single-threaded restatement of the reference's LibDivSufSort on this host, timed on the
    SAME buffer the GPU sorted (one run: ~12 s for 256 MiB), then bit-compared with the GPU SA."""
    import numpy as np
    import oracle
    n = host.size
    t0 = time.perf_counter()
    ref = oracle.divsufsort(host)
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
