"""Batched suffix sorting of independent inputs across the GPUs of one node.

The path shards by INPUT (the many-files bsdiff case: one old file per diff): there is no
data-path collective, because one suffix array never spans devices (every doubling round
would need an all-to-all of ranks over per-link-bound xGMI).

Three layers:

* ``plan_shards``            longest-processing-time-first assignment of inputs to ranks
                             (the same rule as dq_sufsort_hip_batch_i32 uses for devices).
* ``sort_batch_distributed`` one process per GPU under ``torch.distributed``: every rank
                             sorts its share; results are optionally gathered to rank 0
                             (``nccl`` == RCCL over xGMI on the GPU box, ``gloo`` in the CPU
                             tests).  The sorter is injected, so the CPU tests exercise the
                             sharding / gather plumbing without a GPU; on a GPU box it
                             defaults to ``HipSuffixSort``.
* ``diff_many_distributed``  the one exchange step the path has: ONE old file, many new files.  Rank 0 sorts the
                             old file once, text and suffix array are BROADCAST (``ncclBroadcast`` over xGMI under
                             ``nccl``), every rank builds a ``DiffIndex`` on the received buffers and diffs its LPT
                             share of the new files, the patches are gathered to rank 0.  ``Diff.cs:89-90`` is paid
                             once per old file instead of once per (old, new) pair.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np


def plan_shards(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """LPT: inputs sorted by decreasing length, each given to the least-loaded rank.
    Returns, per rank, the list of input indices it owns (in assignment order)."""
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    order = sorted(range(len(lengths)), key=lambda j: (-int(lengths[j]), j))
    shares: List[List[int]] = [[] for _ in range(world_size)]
    load = [0] * world_size
    for j in order:
        r = min(range(world_size), key=lambda d: (load[d], d))
        shares[r].append(j)
        load[r] += int(lengths[j])
    return shares


def sort_batch_local(texts: Sequence, sorter) -> List[np.ndarray]:
    """All inputs on one device, through the provider's ISuffixSort surface."""
    return [sorter.Sort(t) for t in texts]


def sort_batch_distributed(texts: Optional[Sequence], *, sorter_factory: Optional[Callable] = None,
                           gather_to_root: bool = True, group=None) -> Optional[List[np.ndarray]]:
    """Sort a batch across the ranks of an initialised ``torch.distributed`` process group.

    ``texts`` must be given on rank 0 (other ranks may pass ``None``); inputs are
    scattered to their owners, sorted there, and -- if ``gather_to_root`` -- the suffix
    arrays are returned on rank 0 in input order (other ranks return ``None``).
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    # ---- rank 0 announces the plan ----
    if rank == 0:
        arrs = [np.ascontiguousarray(np.frombuffer(memoryview(t).cast("B"), dtype=np.uint8)
                                     if not isinstance(t, np.ndarray) else t) for t in texts]
        lengths = [int(a.size) for a in arrs]
        plan = plan_shards(lengths, world)
        meta = [lengths, plan]
    else:
        arrs, meta = None, None
    box = [meta]
    dist.broadcast_object_list(box, src=0, group=group)
    lengths, plan = box[0]

    # ---- scatter inputs to their owners (point-to-point; rank 0 keeps its own share) ----
    mine = {}
    if rank == 0:
        reqs = []
        for r in range(world):
            for j in plan[r]:
                if r == 0:
                    mine[j] = arrs[j]
                elif lengths[j] > 0:
                    reqs.append(dist.isend(torch.from_numpy(arrs[j]).to(dev), dst=r, group=group))
        for q in reqs:
            q.wait()
    else:
        for j in plan[rank]:
            buf = torch.empty(lengths[j], dtype=torch.uint8, device=dev)
            if lengths[j] > 0:
                dist.recv(buf, src=0, group=group)
            mine[j] = buf

    # ---- sort the local share ----
    if sorter_factory is None:
        from .suffix_sort import HipSuffixSort
        sorter = HipSuffixSort(dev.index if dev.type == "cuda" else -1)
    else:
        sorter = sorter_factory()
    results = {}
    for j in plan[rank]:
        t = mine[j]
        if isinstance(t, torch.Tensor) and not t.is_cuda:
            t = t.numpy()
        sa = sorter.Sort(t)
        results[j] = sa

    if not gather_to_root:
        return [results[j] for j in plan[rank]]

    # ---- gather the suffix arrays to rank 0 ----
    def as_tensor(x):
        if isinstance(x, torch.Tensor):
            return x.to(dev)
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(dev)

    if rank == 0:
        out: List[Optional[np.ndarray]] = [None] * len(lengths)
        for j in plan[0]:
            r0 = results[j]
            out[j] = r0.cpu().numpy() if isinstance(r0, torch.Tensor) else np.asarray(r0)
        for r in range(1, world):
            for j in plan[r]:
                buf = torch.empty(lengths[j], dtype=torch.int32, device=dev)
                if lengths[j] > 0:
                    dist.recv(buf, src=r, group=group)
                out[j] = buf.cpu().numpy()
        return out
    for j in plan[rank]:
        if lengths[j] > 0:
            dist.send(as_tensor(results[j]), dst=0, group=group)
    return None


def diff_many_distributed(old, news: Optional[Sequence], *, sorter_factory: Optional[Callable] = None,
                          index_factory: Optional[Callable] = None, group=None) -> Optional[List[bytes]]:
    """BSDIFF40 patches of many new files against ONE old file across the ranks of a process group.

    ``old`` and ``news`` must be given on rank 0 (other ranks pass ``None``).  Returns, on rank 0, the patches in
    input order (``patch[j]`` turns ``old`` into ``news[j]``; other ranks return ``None``).

    ``sorter_factory()`` -> object with ``Sort(text)`` (default ``HipSuffixSort``);
    ``index_factory(old_bytes, text_tensor, sa_tensor)`` -> object with ``Create(new) -> bytes`` (default
    ``DiffIndex`` on the tensors as they arrived: device tensors under ``nccl``, uploaded once under ``gloo``).
    The CPU tests inject both.
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    on_gpu = backend == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")

    def as_u8(x) -> np.ndarray:
        return x if isinstance(x, np.ndarray) else np.frombuffer(memoryview(x).cast("B"), dtype=np.uint8)

    # ---- plan: new files to ranks by length (the scan loop's cost grows with the new file) ----
    if rank == 0:
        old_np = np.ascontiguousarray(as_u8(old), dtype=np.uint8)
        new_np = [np.ascontiguousarray(as_u8(x), dtype=np.uint8) for x in news]
        lengths = [int(a.size) for a in new_np]
        meta = [int(old_np.size), lengths, plan_shards(lengths, world)]
    else:
        old_np, new_np, meta = None, None, None
    box = [meta]
    dist.broadcast_object_list(box, src=0, group=group)
    n, lengths, plan = box[0]

    # ---- rank 0 sorts the old file; text + suffix array go to every rank in two broadcasts ----
    if rank == 0:
        if sorter_factory is None:
            from .suffix_sort import HipSuffixSort
            sorter = HipSuffixSort(dev.index if on_gpu else -1)
        else:
            sorter = sorter_factory()
        text_t = torch.from_numpy(old_np).to(dev)
        sa = sorter.Sort(text_t if (on_gpu and sorter_factory is None) else old_np)
        sa_t = sa.to(dev) if isinstance(sa, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(sa, dtype=np.int32)).to(dev)
    else:
        text_t = torch.empty(n, dtype=torch.uint8, device=dev)
        sa_t = torch.empty(n, dtype=torch.int32, device=dev)
    if n > 0:
        dist.broadcast(text_t, src=0, group=group)
        dist.broadcast(sa_t, src=0, group=group)
    if rank != 0:
        old_np = text_t.cpu().numpy()                  # the scan loop walks the old file on the host

    # ---- new files to their owners (point-to-point; rank 0 keeps its own share) ----
    mine = {}
    if rank == 0:
        reqs = []
        for r in range(world):
            for j in plan[r]:
                if r == 0:
                    mine[j] = new_np[j]
                elif lengths[j] > 0:
                    reqs.append(dist.isend(torch.from_numpy(new_np[j]).to(dev), dst=r, group=group))
        for q in reqs:
            q.wait()
    else:
        for j in plan[rank]:
            buf = torch.empty(lengths[j], dtype=torch.uint8, device=dev)
            if lengths[j] > 0:
                dist.recv(buf, src=0, group=group)
            mine[j] = buf.cpu().numpy()

    # ---- every rank: one index on the broadcast buffers, then its share of the diffs ----
    if index_factory is None:
        from .bsdiff import DiffIndex
        if not text_t.is_cuda:                         # gloo on a GPU box: upload once
            text_t, sa_t = text_t.cuda(), sa_t.cuda()
        index = DiffIndex(old_np, device_text=text_t, device_sa=sa_t)
    else:
        index = index_factory(old_np, text_t, sa_t)
    patches = {j: index.Create(mine[j]) for j in plan[rank]}
    if hasattr(index, "close"):
        index.close()

    # ---- patches (small: three bzip2 streams) to rank 0 ----
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(patches, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out: List[Optional[bytes]] = [None] * len(lengths)
    for part in gathered:
        for j, p in part.items():
            out[j] = p
    return out
