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
        out = sort_batch_distributed(texts, sorter_factory=OracleSorter)
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
