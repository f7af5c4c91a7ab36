"""numpy model of the GPU prefix-doubling pipeline (test infrastructure).

It mirrors, step for step, what deltaq_amd/csrc does on the device -- 8-byte
big-endian zero-padded initial keys, stable sort, head marking, group-start
ranks, compaction of non-singleton groups, composite (rank, key2) keys with the
"past the end" rule, rebucketing -- so the *algorithm* can be validated on the
CPU against the oracle before any kernel runs.  Not used by the product.
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


def suffix_array(T: np.ndarray, trace: list | None = None) -> np.ndarray:
    T = np.ascontiguousarray(T, dtype=np.uint8)
    n = T.size
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    key = initial_keys(T)
    order = np.argsort(key, kind="stable")
    ck = key[order]
    s = order.astype(np.int64)

    SA = np.empty(n, dtype=np.int64)
    ISA = np.empty(n, dtype=np.int64)
    # initial "apply": one group [0, n)
    r = np.zeros(n, dtype=np.int64)
    h = 8
    first = True
    while True:
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
        ISA[s] = rnew
        nxt = np.ones(m, dtype=bool)
        nxt[:-1] = newhead[1:]
        active = ~(newhead & nxt)
        s = s[active]
        r = rnew[active]
        if trace is not None:
            trace.append((h, int(s.size)))
        if s.size == 0:
            break
        # gather key2 with the past-the-end rule: shorter suffix first
        q = s + h
        k2 = np.where(q < n, ISA[np.minimum(q, n - 1)] + h, n - 1 - s)
        # composite sort (stable by (r, k2)); active list is already sorted by r
        o = np.lexsort((k2, r))
        s, r, ck = s[o], r[o], k2[o].astype(np.uint64)
        h *= 2
        first = False
    return SA
