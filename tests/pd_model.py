"""numpy model of the GPU prefix-doubling pipeline (test infrastructure).

It mirrors, step for step, what deltaq_amd/csrc does on the device -- kb-byte big-endian
zero-padded round-0 keys, sort, head marking, group-start ranks, compaction of non-singleton
groups, the sparse finishing path (key extension from the text with a (bytes, length) key),
the switch to dense doubling (ISA from SA + ranks of the still-tied suffixes), composite
(rank, key2) keys with the "past the end" rule, rebucketing -- so the *algorithm* can be
validated on the CPU against the oracle before any kernel runs.  Not used by the product.
"""
from __future__ import annotations

import numpy as np


def initial_keys(T: np.ndarray, kbytes: int = 8) -> np.ndarray:
    n = T.size
    P = np.concatenate([T, np.zeros(kbytes, dtype=np.uint8)]).astype(np.uint64)
    key = np.zeros(n, dtype=np.uint64)
    for b in range(kbytes):
        key = (key << np.uint64(8)) | P[b:b + n]
    return key


def _rebucket(s, r, ck, SA, ISA, write_isa):
    """seg_reduce/scan/apply: s, r, ck sorted by (r, ck).  Returns the still-active (s, r)."""
    m = s.size
    j = np.arange(m, dtype=np.int64)
    newhead = np.ones(m, dtype=bool)
    newhead[1:] = (ck[1:] != ck[:-1]) | (r[1:] != r[:-1])
    grouphead = np.ones(m, dtype=bool)
    grouphead[1:] = r[1:] != r[:-1]
    nh = np.maximum.accumulate(np.where(newhead, j, -1))
    gh = np.maximum.accumulate(np.where(grouphead, j, -1))
    p = r + (j - gh)
    rnew = r + (nh - gh)
    SA[p] = s
    if write_isa:
        ISA[s] = rnew
    nxt = np.ones(m, dtype=bool)
    nxt[:-1] = newhead[1:]
    active = ~(newhead & nxt)
    return s[active], rnew[active]


def suffix_array(T: np.ndarray, trace: list | None = None, kbytes: int = 8,
                 sparse: bool | None = None, ebytes: int = 4, sparse_rounds: int = 3,
                 keys: np.ndarray | None = None) -> np.ndarray:
    """keys: round-0 keys from elsewhere (the coded keys of dq_coded_keys.h) -- any keys that are monotone in the
    suffix order and whose equality implies `kbytes` equal leading characters."""
    T = np.ascontiguousarray(T, dtype=np.uint8)
    n = T.size
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    key = initial_keys(T, kbytes) if keys is None else np.ascontiguousarray(keys, dtype=np.uint64)
    order = np.argsort(key, kind="stable")[::1]
    # round 0 need not be stable with respect to the text order: shuffle inside equal keys
    ck = key[order]
    s = order.astype(np.int64)
    SA = np.full(n, -1, dtype=np.int64)
    ISA = np.full(n, -1, dtype=np.int64)

    r = np.zeros(n, dtype=np.int64)
    h = kbytes
    # initial rebucket; decide sparse / dense from the number of ties
    m_probe = _rebucket(s, r, ck, SA.copy(), ISA.copy(), False)[0].size
    if sparse is None:
        sparse = m_probe * 32 <= n
    s, r = _rebucket(s, r, ck, SA, ISA, not sparse)
    if trace is not None:
        trace.append(("init", h, int(s.size), "sparse" if sparse else "dense"))

    Tp = np.concatenate([T, np.zeros(ebytes + 1, dtype=np.uint8)]).astype(np.uint64)
    if sparse:
        for _ in range(sparse_rounds):
            if s.size == 0:
                break
            q = s + h
            ln = np.clip(n - q, 0, ebytes)
            byts = np.zeros(s.size, dtype=np.uint64)
            for b in range(ebytes):
                qb = np.minimum(q + b, n)          # index n.. reads the zero pad
                byts = (byts << np.uint64(8)) | np.where(b < ln, Tp[np.minimum(qb, Tp.size - 1)], 0).astype(np.uint64)
            k2 = (byts << np.uint64(3)) | ln.astype(np.uint64)
            o = np.lexsort((k2, r))
            s, r, k2 = s[o], r[o], k2[o]
            s, r = _rebucket(s, r, k2, SA, ISA, False)
            h += ebytes
            if trace is not None:
                trace.append(("sparse", h, int(s.size)))
        if s.size:
            ISA[SA] = np.arange(n, dtype=np.int64)      # isa_from_sa
            ISA[s] = r                                   # isa_scatter

    while s.size:
        q = s + h
        k2 = np.where(q < n, ISA[np.minimum(q, n - 1)] + h, n - 1 - s).astype(np.uint64)
        o = np.lexsort((k2, r))
        s, r, k2 = s[o], r[o], k2[o]
        s, r = _rebucket(s, r, k2, SA, ISA, True)
        h *= 2
        if trace is not None:
            trace.append(("dense", h, int(s.size)))
    return SA
