"""Batched suffix sorting of independent inputs across the GPUs of one node.

The path shards by INPUT (the many-files bsdiff case: one old file per diff): there is no
data-path collective, because one suffix array never spans devices (every doubling round
would need an all-to-all of ranks over per-link-bound xGMI).

Three layers:

* ``plan_shards``            longest-processing-time-first assignment of inputs to ranks
                             (the same rule as dq_sufsort_hip_batch_i32 uses for devices).
* ``sort_batch_distributed`` one process per GPU under ``torch.distributed``: every rank is sent its LPT share
                             (``dist.scatter`` of padded shares: no rank holds more than its own), sorts
                             it, the suffix arrays come back to rank 0 with one padded gather
                             (``nccl`` == RCCL over xGMI on the GPU box, ``gloo`` in the CPU tests).
                             Collectives only: no point-to-point pairs.  The sorter is injected, so the
                             CPU tests exercise the plumbing without a GPU; on a GPU box it defaults to
                             ``HipSuffixSort``.
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


def _announce(rank, arrs, world, group, extra=None):
    """rank 0 -> everybody: the lengths of the inputs, the LPT plan, and anything else that is small."""
    import torch.distributed as dist
    if rank == 0:
        lengths = [int(a.size) for a in arrs]
        meta = [lengths, plan_shards(lengths, world), extra]
    else:
        meta = None
    box = [meta]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


# rank 0 stages at most this many bytes per rank and scatter step (the shares travel in rows of this width)
_SCATTER_ROW_BYTES = 128 << 20


def _deal_out(rank, arrs, lengths, plan, dev, group, row_bytes: int = _SCATTER_ROW_BYTES):
    """The inputs to their owners: every rank receives ITS OWN share and nothing else.  Rank 0 lays each rank's
    share out input after input, pads the shares to the longest one and scatters them (`dist.scatter`: grouped
    sends / receives inside RCCL over xGMI under nccl) in rows of at most `row_bytes` per rank, so that a rank holds
    its share (128 x 16 MiB over 8 ranks: 256 MiB each, where one broadcast of the whole batch put 2 GiB on every
    device and moved world times the bytes) and rank 0 stages world x row_bytes at a time.
    Returns {input index: uint8 tensor on `dev`} for this rank's share."""
    import torch
    import torch.distributed as dist
    world = len(plan)
    share_len = [sum(lengths[j] for j in share) for share in plan]
    cap = max(share_len) if share_len else 0
    mine_flat = torch.empty(share_len[rank], dtype=torch.uint8, device=dev)
    flats = None
    if rank == 0:
        flats = []
        for share in plan:
            flat = np.empty(sum(lengths[j] for j in share), dtype=np.uint8)
            pos = 0
            for j in share:
                flat[pos:pos + lengths[j]] = arrs[j]
                pos += lengths[j]
            flats.append(flat)
    for lo in range(0, cap, max(1, row_bytes)):
        w = min(row_bytes, cap - lo)
        recv = torch.empty(w, dtype=torch.uint8, device=dev)
        rows = None
        if rank == 0:
            rows = []
            for r in range(world):
                row = np.zeros(w, dtype=np.uint8)
                part = flats[r][lo:lo + w]
                row[:part.size] = part
                rows.append(torch.from_numpy(row).to(dev))
        dist.scatter(recv, rows, src=0, group=group)
        take = max(0, min(w, share_len[rank] - lo))
        if take:
            mine_flat[lo:lo + take] = recv[:take]
    mine, pos = {}, 0
    for j in plan[rank]:
        mine[j] = mine_flat[pos:pos + lengths[j]]
        pos += lengths[j]
    return mine


def _all_ok(err, dev, group) -> None:
    """Every rank reports whether its local work succeeded BEFORE the result collective: a rank that failed would
    otherwise leave the others waiting in it.  Raises on every rank if any failed."""
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()) != 0:
        raise RuntimeError(f"a rank failed in its share of the batch (this rank: {err!r})")


def _collect(rank, world, plan, sizes, parts, dtype, dev, group):
    """Variable-length results to rank 0 with ONE collective: every rank concatenates its results (in plan order),
    pads to the longest share and takes part in `dist.gather` (grouped sends / receives inside RCCL).
    sizes[j] = elements of result j; parts = {j: tensor} of this rank.  Returns {j: tensor on dev} on rank 0."""
    import torch
    import torch.distributed as dist
    share_len = [sum(sizes[j] for j in share) for share in plan]
    cap = max(max(share_len), 1)
    mine = torch.zeros(cap, dtype=dtype, device=dev)
    pos = 0
    for j in plan[rank]:
        if sizes[j]:
            mine[pos:pos + sizes[j]] = parts[j].to(dev).reshape(-1)
        pos += sizes[j]
    bins = [torch.empty(cap, dtype=dtype, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, bins, dst=0, group=group)
    if rank != 0:
        return None
    out = {}
    for r, share in enumerate(plan):
        pos = 0
        for j in share:
            out[j] = bins[r][pos:pos + sizes[j]]
            pos += sizes[j]
    return out


def sort_batch_distributed(texts: Optional[Sequence], *, sorter_factory: Optional[Callable] = None,
                           gather_to_root: bool = True, group=None) -> Optional[List[np.ndarray]]:
    """Sort a batch across the ranks of an initialised ``torch.distributed`` process group.

    ``texts`` must be given on rank 0 (other ranks may pass ``None``); inputs are dealt out to their owners (a
    scatter of the shares), sorted there, and -- if ``gather_to_root`` -- the suffix arrays are returned on rank 0 in input
    order (one gather; other ranks return ``None``).
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    arrs = None
    if rank == 0:
        arrs = [np.ascontiguousarray(np.frombuffer(memoryview(t).cast("B"), dtype=np.uint8)
                                     if not isinstance(t, np.ndarray) else t) for t in texts]
    lengths, plan, _ = _announce(rank, arrs, world, group)
    mine = _deal_out(rank, arrs, lengths, plan, dev, group)

    # ---- sort the local share ----
    if sorter_factory is None:
        from .suffix_sort import HipSuffixSort
        sorter = HipSuffixSort(dev.index if dev.type == "cuda" else -1)
    else:
        sorter = sorter_factory()
    results, err = {}, None
    try:
        for j in plan[rank]:
            t = mine[j]
            if not t.is_cuda:
                t = t.numpy()
            elif lengths[j] == 0:
                t = np.zeros(0, np.uint8)
            results[j] = sorter.Sort(t.contiguous() if isinstance(t, torch.Tensor) else t)
    except Exception as e:                      # noqa: BLE001 -- reported to every rank below
        err = e
    _all_ok(err, dev, group)

    if not gather_to_root:
        return [results[j] for j in plan[rank]]

    def as_tensor(x):
        if isinstance(x, torch.Tensor):
            return x.to(torch.int32)
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32))

    got = _collect(rank, world, plan, lengths, {j: as_tensor(v) for j, v in results.items()}, torch.int32, dev, group)
    if rank != 0:
        return None
    return [got[j].cpu().numpy() for j in range(len(lengths))]


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
    old_np = new_np = None
    if rank == 0:
        old_np = np.ascontiguousarray(as_u8(old), dtype=np.uint8)
        new_np = [np.ascontiguousarray(as_u8(x), dtype=np.uint8) for x in news]
    lengths, plan, n = _announce(rank, new_np, world, group, extra=int(old_np.size) if rank == 0 else None)

    # ---- rank 0 sorts the old file; text + suffix array go to every rank in two broadcasts ----
    # (a sort that fails on rank 0 -- out of memory, a device error -- must not leave the others waiting in the broadcast)
    err0 = None
    text_t = sa_t = None
    try:
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
    except Exception as e:                      # noqa: BLE001 -- reported to every rank below
        err0 = e
    _all_ok(err0, dev, group)
    if n > 0:
        dist.broadcast(text_t, src=0, group=group)
        dist.broadcast(sa_t, src=0, group=group)
    if rank != 0:
        old_np = text_t.cpu().numpy()                  # the scan loop walks the old file on the host

    # ---- new files to their owners (a scatter of the shares) ----
    mine = {j: t.cpu().numpy() for j, t in _deal_out(rank, new_np, lengths, plan, dev, group).items()}

    # ---- every rank: one index on the broadcast buffers, then its share of the diffs ----
    patches, err = {}, None
    try:
        if index_factory is None:
            from .bsdiff import DiffIndex
            if not text_t.is_cuda:                     # gloo on a GPU box: upload once
                text_t, sa_t = text_t.cuda(), sa_t.cuda()
            index = DiffIndex(old_np, device_text=text_t, device_sa=sa_t)
        else:
            index = index_factory(old_np, text_t, sa_t)
        patches = {j: index.Create(mine[j]) for j in plan[rank]}
        if hasattr(index, "close"):
            index.close()
    except Exception as e:                      # noqa: BLE001 -- reported to every rank below
        err = e
    _all_ok(err, dev, group)

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
