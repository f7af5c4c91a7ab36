"""Tests of the N>1 path: LPT sharding and the torch.distributed scatter / sort / gather plumbing.
CPU (world_size 2, gloo): the sorter is injected (the oracle, used here as the checker's stand-in) because the
product has no CPU sort path.  GPU box (-m gpu): the same plumbing with HipSuffixSort in every rank."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_plan_shards_is_lpt_and_complete():
    from deltaq_amd.batch import plan_shards
    lens = [16, 3, 9, 9, 1, 0, 20, 7]
    plan = plan_shards(lens, 3)
    assert sorted(j for share in plan for j in share) == list(range(len(lens)))
    loads = [sum(lens[j] for j in share) for share in plan]
    assert max(loads) - min(loads) <= max(lens)            # LPT bound
    assert plan_shards([5] * 128, 8) == [[r + 8 * k for k in range(16)] for r in range(8)]   # round-robin on ties
    assert plan_shards([], 2) == [[], []]
    with pytest.raises(ValueError):
        plan_shards([1], 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle
    from deltaq_amd.batch import sort_batch_distributed

    class OracleSorter:                      # stands in for HipSuffixSort on the CPU
        def Sort(self, text):
            return oracle.divsufsort(np.asarray(text, dtype=np.uint8))

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        texts = None
        if rank == 0:
            texts = [oracle.gen_uniform(5000 + 777 * j, 0x5EED0500 + j) for j in range(5)]
            texts += [np.zeros(0, np.uint8), oracle.gen_enwik_like(20000, 3, 2048), b"banana"]
        # (rows of 3000 bytes: the shares travel in several scatter steps, the last one ragged)
        from deltaq_amd import batch as batch_mod
        batch_mod._SCATTER_ROW_BYTES = 3000
        seen = []
        deal = batch_mod._deal_out
        def spy(rank_, arrs, lengths, plan, dev, group, row_bytes=None):
            mine = deal(rank_, arrs, lengths, plan, dev, group, row_bytes=batch_mod._SCATTER_ROW_BYTES)
            seen.append((sum(int(t.numel()) for t in mine.values()), sum(lengths[j] for j in plan[rank_]), sum(lengths)))
            return mine
        batch_mod._deal_out = spy
        out = sort_batch_distributed(texts, sorter_factory=OracleSorter)
        # a rank holds its own share, not the batch
        assert len(seen) == 1 and seen[0][0] == seen[0][1] < seen[0][2], seen
        if rank == 0:
            ok = all(np.array_equal(o, oracle.divsufsort(np.frombuffer(bytes(t), np.uint8) if isinstance(t, bytes) else t))
                     for o, t in zip(out, texts))
            q.put(("ok" if ok else "mismatch", len(out)))
        else:
            q.put(("ok" if out is None else "non-root returned data", 0))
    finally:
        dist.destroy_process_group()


def test_scatter_sort_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] == "ok" for r in res), res
    assert max(r[1] for r in res) == 8


def _packed(y):
    import struct
    b = bytearray(struct.pack("<Q", abs(y)))
    if y < 0:
        b[7] |= 0x80
    return bytes(b)


def _diff_worker(rank, world, port, q):
    """diff_many_distributed on the CPU: the oracle stands in for the sorter and for the scan loop (the product has
    no CPU path); what is under test is the plan, the broadcast of (text, suffix array), the scatter of the new
    files and the gather of the patches."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bz2
    import torch.distributed as dist
    import oracle
    from deltaq_amd.batch import diff_many_distributed

    class OracleSorter:
        def Sort(self, text):
            return oracle.divsufsort(np.asarray(text, dtype=np.uint8))

    class OracleIndex:                       # Diff.Create from the BROADCAST suffix array: a wrong one gives wrong patches
        def __init__(self, old, text_t, sa_t):
            self.old = np.asarray(old, dtype=np.uint8)
            assert np.array_equal(text_t.numpy(), self.old)
            self.sa = sa_t.numpy().astype(np.int32)

        def Create(self, new):
            new = np.asarray(new, dtype=np.uint8)
            ctrl, diff, extra, _ = oracle.bsdiff_scan(self.old, self.sa, new)
            c = b"".join(_packed(int(v)) for v in ctrl.reshape(-1))
            zc, zd, ze = bz2.compress(c), bz2.compress(diff.tobytes()), bz2.compress(extra.tobytes())
            return b"BSDIFF40" + _packed(len(zc)) + _packed(len(zd)) + _packed(new.size) + zc + zd + ze

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        old = news = None
        if rank == 0:
            old = oracle.gen_enwik_like(30000, 7, 2048)
            news = []
            for j in range(7):
                x = old.copy()
                x[1000 * j:1000 * j + 50] = oracle.gen_uniform(50, j)
                news.append(np.concatenate([x[:20000 - 1500 * j], oracle.gen_uniform(100 * j, 50 + j), x[20000 - 1500 * j:]]))
            news += [np.zeros(0, np.uint8), old.copy(), oracle.gen_uniform(500, 99)]
        out = diff_many_distributed(old, news, sorter_factory=OracleSorter, index_factory=OracleIndex)
        if rank == 0:
            from deltaq_amd import Patch        # Patch.Apply is host code: it runs without a GPU
            ok = len(out) == len(news) and all(Patch.Apply(old, p) == x.tobytes() for p, x in zip(out, news))
            q.put(("ok" if ok else "mismatch", len(out)))
        else:
            q.put(("ok" if out is None else "non-root returned data", 0))
    finally:
        dist.destroy_process_group()


def test_one_old_many_new_world_size_2_gloo(backend_lib):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_diff_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] == "ok" for r in res), res
    assert max(r[1] for r in res) == 10


def _gpu_worker(rank, world, port, q, backend):
    """The same plumbing with the PRODUCT's sorter: every rank is its own process with its own HIP context
    (the box has one GPU: with gloo both ranks use device 0; with nccl, world size 1)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import oracle
    from deltaq_amd import HipSuffixSort
    from deltaq_amd.batch import sort_batch_distributed

    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        texts = None
        if rank == 0:
            texts = [oracle.gen_uniform(300_000 + 7777 * j, 0x5EED0500 + j) for j in range(6)]
            texts += [np.zeros(0, np.uint8), oracle.gen_enwik_like(200_000, 3, 8192), b"banana"]
        out = sort_batch_distributed(texts, sorter_factory=None if backend == "nccl" else (lambda: HipSuffixSort(0)))
        if rank == 0:
            ok = all(np.array_equal(np.asarray(o), oracle.divsufsort(np.frombuffer(bytes(t), np.uint8) if isinstance(t, bytes) else t))
                     for o, t in zip(out, texts))
            q.put(("ok" if ok else "mismatch", len(out)))
        else:
            q.put(("ok" if out is None else "non-root returned data", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_scatter_sort_gather_with_the_hip_sorter(backend, world):
    """sort_batch_distributed + HipSuffixSort in separate processes: gloo with two ranks sharing the GPU, and the
    RCCL backend (process group initialisation, device tensors end to end) with the one rank a 1-GPU box allows."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] == "ok" for r in res), res
    assert max(r[1] for r in res) == 9


def _gpu_diff_worker(rank, world, port, q, backend):
    """diff_many_distributed with the PRODUCT's pieces in every rank: rank 0 sorts on the device, text and suffix
    array are broadcast (RCCL under nccl), every rank builds a DiffIndex on the received buffers."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import oracle
    from deltaq_amd import Diff, Patch
    from deltaq_amd.batch import diff_many_distributed

    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        old = news = None
        if rank == 0:
            old = oracle.gen_enwik_like(1_500_000, 7, 16384)
            news = []
            for j in range(6):
                x = old.copy()
                for e in range(40):
                    a = (j * 7919 + e * 36_000) % (old.size - 100)
                    x[a:a + 8] = oracle.gen_uniform(8, 100 * j + e)
                news.append(np.concatenate([x[:700_000 - 9000 * j], oracle.gen_uniform(300 * j, 50 + j), x[700_000 - 9000 * j:]]))
            news += [np.zeros(0, np.uint8), old.copy(), oracle.gen_uniform(70_000, 99)]
        out = diff_many_distributed(old, news)
        if rank == 0:
            ok = len(out) == len(news)
            for p, x in zip(out, news):
                ok = ok and Patch.Apply(old, p) == x.tobytes()
                ok = ok and p == Diff.CreateBytes(old, x, 0)             # the patch Diff.Create writes, byte for byte
            q.put(("ok" if ok else "mismatch", len(out)))
        else:
            q.put(("ok" if out is None else "non-root returned data", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_one_old_many_new_with_the_hip_pieces(backend, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_diff_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] == "ok" for r in res), res
    assert max(r[1] for r in res) == 9


@pytest.mark.gpu
def test_bench_line_of_two_ranks_sharing_the_gpu():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), here with both ranks on
    cuda:0 over gloo: the line must carry the N > 1 records -- configs[4] LPT-sharded 64 / 64, one run of the scatter /
    sort / gather layer and one of the one-old-many-new layer -- without an error key."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in [k for k in env if k.startswith("DQ_")]:
        env.pop(k)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--share-gpu", "--backend", "gloo", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "weak"
    assert rec["roofline"]["frac_of_6p3"] > rec["roofline"]["frac"] > 0
    b = rec["batch"]
    assert b["sharding"]["buffers_per_rank"] == [64, 64]
    assert b["sufcheck_first_buffers"] is True
    assert len(b["host_copy_GBps_per_rank"]) == 2
    assert len(b["sort_ms_per_rank"]) == 2 and all(x > 0 for x in b["sort_ms_per_rank"])
    ssg = b["rccl_scatter_sort_gather"]
    assert "error" not in ssg and ssg["sufcheck"] is True and ssg["buffers"] == 4, ssg
    many = b["one_old_many_new"]
    assert "error" not in many and many["patches_apply"] is True and many["new_files"] == 8, many


@pytest.mark.gpu
def test_bench_line_of_eight_ranks_sharing_the_gpu():
    """The driver's first real 8-GPU run must not be the first time world 8 executes: bench.py as it launches it for N = 8,
    all ranks on cuda:0 over gloo, at small sizes (16 MiB headline buffers, a 32 x 4 MiB batch, one new file per rank) --
    scatter rows, gather, broadcast, the LPT plan, `devices_per_rank`, the per-rank stage times and the NUMA fields for
    eight ranks."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in [k for k in env if k.startswith("DQ_")]:
        env.pop(k)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--size-mib", "16", "--batch-count", "32", "--batch-mib", "4", "--many-new-per-rank", "1",
           "--share-gpu", "--backend", "gloo", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["gpus_flag"] == 8 and rec["scaling"] == "weak"
    assert rec["devices_per_rank"] == [0] * 8 and len(rec["numa_node_per_rank"]) == 8
    b = rec["batch"]
    assert b["workload"].startswith("NOT configs[4]")             # (the named workload is 128 x 16 MiB)
    assert b["sharding"]["buffers_per_rank"] == [4] * 8
    assert b["sufcheck_first_buffers"] is True
    for key in ("host_copy_GBps_per_rank", "wall_ms_per_rank", "sort_ms_per_rank", "copy_in_ms_per_rank", "copy_out_ms_per_rank"):
        assert len(b[key]) == 8 and all(x > 0 for x in b[key]), (key, b[key])
    ssg = b["rccl_scatter_sort_gather"]
    assert "error" not in ssg and ssg["sufcheck"] is True and ssg["buffers"] == 16, ssg
    many = b["one_old_many_new"]
    assert "error" not in many and many["patches_apply"] is True and many["new_files"] == 8, many


def _bench_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in [k for k in env if k.startswith("DQ_") or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")]:
        env.pop(k)
    return env


def test_bench_refuses_a_launcher_that_disagrees_with_gpus_flag():
    """`--gpus N` is not decoration: a world size that differs from it is an error, not a 1-GPU number (round-4 verdict:
    the flag was parsed and never read, so `bench.py --gpus 8` without a launcher measured one GPU with rc 0)."""
    import subprocess
    env = dict(_bench_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env,
                       timeout=300, cwd=ROOT)
    assert p.returncode == 2 and "disagree" in p.stderr, (p.returncode, p.stderr[-500:])
    assert '"metric"' not in p.stdout


def test_bench_without_a_launcher_starts_its_own_ranks():
    """No GPU here: what can be checked is that `python bench.py --gpus 2` becomes a torch.distributed.run of two ranks
    (each of which then refuses to run without a GPU) and that the failure comes back as the exit status -- never a
    line with n_gpus 1."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: test_bench_without_a_launcher_runs_two_ranks covers it")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=_bench_env(), timeout=600, cwd=ROOT)
    assert p.returncode != 0
    assert "starting the ranks" in p.stderr and "--nproc-per-node 2" in p.stderr, p.stderr[-1500:]
    assert p.stderr.count("bench.py needs a GPU") >= 1, p.stderr[-1500:]
    assert '"metric"' not in p.stdout


@pytest.mark.gpu
def test_bench_without_a_launcher_runs_two_ranks():
    """`python bench.py --gpus 2 --share-gpu --backend gloo` with NO launcher around it: bench.py starts the two ranks
    itself and the line says n_gpus 2, with configs[4] dealt 64 / 64."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--share-gpu", "--backend", "gloo", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=_bench_env(), timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["gpus_flag"] == 2 and rec["devices_per_rank"] == [0, 0]
    assert rec["batch"]["sharding"]["buffers_per_rank"] == [64, 64]
