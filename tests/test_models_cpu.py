"""numpy models of the device algorithms added on top of plain prefix doubling, checked against direct
definitions.  They pin the LOGIC (what the kernels in dq_ties.h / dq_isa_pairs.h / dq_small_groups.h are
specified to compute) on the CPU; the kernels themselves are checked bit-exact on the GPU (test_gpu_parity.py)."""
import numpy as np
import pytest


def sorted_keys(rng, n, distinct):
    return np.sort(rng.integers(0, distinct, size=n, dtype=np.uint64))


# ---------------------------------------------------------------- tie bits of the last digit pass (dq_ties.h)
def tie_bits_model(keys_in, tile, top_shift):
    """keys_in: the last pass's INPUT order (sorted by the low digits, arbitrary top digit).  Emulates the pass:
    per tile, a stable split by the top digit; E bits for neighbours inside one (tile, digit) run; the first /
    last key of every run in a seam table; then the seam decision.  Returns (sorted keys, E)."""
    n = keys_in.size
    digit = (keys_in >> np.uint64(top_shift)).astype(np.int64)
    ntiles = (n + tile - 1) // tile
    counts = np.zeros((ntiles, 256), np.int64)
    for t in range(ntiles):
        counts[t] = np.bincount(digit[t * tile:(t + 1) * tile], minlength=256)
    dig_tot = counts.sum(axis=0)
    dig_off = np.concatenate([[0], np.cumsum(dig_tot)[:-1]])
    excl = np.cumsum(counts, axis=0) - counts                      # keys with digit d in tiles before t
    out = np.zeros(n, np.uint64)
    E = np.zeros(n, bool)
    first = np.full((ntiles, 256), -1, np.int64); last = np.full((ntiles, 256), -1, np.int64)
    for t in range(ntiles):
        k = keys_in[t * tile:(t + 1) * tile]; d = digit[t * tile:(t + 1) * tile]
        order = np.argsort(d, kind="stable")
        ks, ds = k[order], d[order]
        pos = np.arange(ks.size)
        run_start = np.concatenate([[0], np.cumsum(counts[t])[:-1]])
        o = dig_off[ds] + excl[t, ds] + (pos - run_start[ds])
        out[o] = ks
        same_run = np.concatenate([[False], ds[1:] == ds[:-1]])
        eq = np.concatenate([[False], ks[1:] == ks[:-1]]) & same_run
        E[o[eq]] = True
        for dd in np.unique(ds):
            idx = np.nonzero(ds == dd)[0]
            first[t, dd] = idx[0]; last[t, dd] = idx[-1]
            first[t, dd] = int(ks[idx[0]]); last[t, dd] = int(ks[idx[-1]])
    # seam kernel: first key of run (t, d) against the last key of the nearest earlier tile that has digit d
    for t in range(1, ntiles):
        for dd in range(256):
            if counts[t, dd] == 0:
                continue
            tp = t - 1
            while tp >= 0 and counts[tp, dd] == 0:
                tp -= 1
            if tp >= 0 and last[tp, dd] == first[t, dd]:
                E[dig_off[dd] + excl[t, dd]] = True
    return out, E


def collect_model(E):
    """tie_collect_kernel: members of groups of size > 1 as (rank = position of the first member, position)."""
    n = E.size
    active = E | np.concatenate([E[1:], [False]])
    head = np.maximum.accumulate(np.where(~E, np.arange(n), 0))
    p = np.nonzero(active)[0]
    return head[p], p


@pytest.mark.parametrize("n,distinct,tile", [(5000, 3000, 512), (4096, 40, 256), (3000, 1, 128), (7777, 10**9, 1024),
                                              (2048, 600, 2048), (1, 5, 64), (257, 2, 64)])
def test_tie_bits_and_seams_equal_the_direct_definition(n, distinct, tile):
    rng = np.random.default_rng(n + distinct)
    keys = rng.integers(0, distinct, size=n, dtype=np.uint64) | (rng.integers(0, 7, size=n, dtype=np.uint64) << np.uint64(56))
    low_sorted = keys[np.argsort(keys & np.uint64((1 << 56) - 1), kind="stable")]          # LSD: low digits done
    out, E = tie_bits_model(low_sorted, tile, 56)
    assert np.array_equal(out, np.sort(keys))
    direct = np.concatenate([[False], out[1:] == out[:-1]])
    assert np.array_equal(E, direct)
    rank, pos = collect_model(E)
    # every tied position appears once, with the position of its group's first member as rank
    _, first_idx, inv, cnt = np.unique(out, return_index=True, return_inverse=True, return_counts=True)
    tied = np.nonzero(cnt[inv] > 1)[0]
    assert np.array_equal(pos, tied)
    assert np.array_equal(rank, first_idx[inv][tied])


# ---------------------------------------------------------------- suffix-binned first ISA (dq_isa_pairs.h)
@pytest.mark.parametrize("n", [1 << 16, (1 << 16) + 1, 100_003, 1 << 17])
def test_binned_words_put_consecutive_suffixes_in_consecutive_positions(n):
    rng = np.random.default_rng(n)
    ib = int(n - 1).bit_length()
    sa = rng.permutation(n).astype(np.uint64)
    rank = np.sort(rng.integers(0, n, size=n)).astype(np.uint64)       # any non-decreasing rank array
    words = (rank << np.uint64(ib)) | sa
    # closed-form digit offsets of the two binning passes: every suffix occurs exactly once
    for sh in (ib - 16, ib - 8):
        unit = 1 << sh
        full, rem = n >> (sh + 8), n & ((unit << 8) - 1)
        cnt = np.array([full * unit + min(max(rem - d * unit, 0), unit) for d in range(256)])
        d = ((words & np.uint64((1 << ib) - 1)) >> np.uint64(sh)) & np.uint64(255)
        assert np.array_equal(cnt, np.bincount(d.astype(np.int64), minlength=256))
        words = words[np.argsort(d, kind="stable")]
    suf = (words & np.uint64((1 << ib) - 1)).astype(np.int64)
    span = 4096
    isa = np.zeros(n, np.int64)
    for base in range(0, n, span):
        s = suf[base:base + span]
        assert s.min() == base and s.max() == min(base + span, n) - 1          # the LDS image covers exactly its span
        image = np.zeros(span, np.int64)
        image[s - base] = (words[base:base + span] >> np.uint64(ib)).astype(np.int64)
        isa[base:base + s.size] = image[:s.size]
    direct = np.zeros(n, np.int64); direct[sa.astype(np.int64)] = rank.astype(np.int64)
    assert np.array_equal(isa, direct)


# ---------------------------------------------------------------- place inside a small group (dq_small_groups.h)
def test_place_by_counting_is_a_stable_sort_and_finds_the_ties():
    rng = np.random.default_rng(5)
    for g in range(1, 33):
        k2 = rng.integers(0, max(2, g // 2), size=g)
        less = np.array([(k2 < x).sum() for x in k2])
        eq_before = np.array([(k2[:i] == k2[i]).sum() for i in range(g)])
        place = less + eq_before
        assert sorted(place.tolist()) == list(range(g))
        assert np.array_equal(k2[np.argsort(place)], np.sort(k2))
        tied = np.array([(k2 == x).sum() > 1 for x in k2])
        new_rank = less                                             # rank offset of the subgroup
        for i in range(g):
            assert tied[i] == ((new_rank == new_rank[i]).sum() > 1)


def pair_chain_model(T, sa, h, maxg=4, jumps=4):
    """dq_pair_chains.h in plain Python: groups = suffixes agreeing on h characters (ranks = group starts, as the
    doubling rounds have them); every group of <= maxg members becomes its pairs (d, x); records sorted; a record
    links to the next one if that is (d, x+1); a chain end decides by the ranks of x+1 / y+1, by a far link, by
    stepping h characters, or stays blocked.  Returns {(x, y): 1 (x first) | 2 (y first) | 3 (blocked)}."""
    n = T.size
    b = T.tobytes()
    rank = np.zeros(n, np.int64)                      # group rank of every suffix at depth h (padding: shorter first)
    start = 0
    for p in range(1, n + 1):
        if p == n or b[sa[p]:sa[p] + h] != b[sa[start]:sa[start] + h]:
            rank[sa[start:p]] = start
            start = p
    groups = {}
    for s in range(n):
        groups.setdefault(int(rank[s]), []).append(s)
    recs = []
    for g in groups.values():
        if 2 <= len(g) <= maxg:
            g = sorted(g)
            recs += [(y - x, x) for i, x in enumerate(g) for y in g[i + 1:]]
    recs.sort()
    index = {r: i for i, r in enumerate(recs)}
    status, far = [0] * len(recs), [0] * len(recs)
    for i, (d, x) in enumerate(recs):
        if i + 1 < len(recs) and recs[i + 1] == (d, x + 1):
            continue                                   # link
        xx, st = x + 1, 3
        for _ in range(jumps + 1):
            yy = xx + d
            if yy >= n:
                st = 2
                break
            if rank[xx] != rank[yy]:
                st = 1 if rank[xx] < rank[yy] else 2
                break
            if (d, xx) in index:
                st, far[i] = 4, index[(d, xx)]
                break
            xx += h
        status[i] = st
    end = [0] * len(recs)                              # first chain end at or after i
    for i in range(len(recs) - 1, -1, -1):
        end[i] = i if status[i] else end[i + 1]
    def resolve(i, depth=0):
        e = end[i]
        if status[e] != 4:
            return status[e]
        return resolve(far[e], depth + 1) if depth < 64 else 3
    return {(x, x + d): resolve(i) for i, (d, x) in enumerate(recs)}


def test_pair_chains_decide_like_the_suffix_array(oracle_mod):
    rng = np.random.default_rng(31)
    x = oracle_mod.gen_uniform(3000, 1) & 7
    texts = [
        np.concatenate([x, rng.integers(0, 8, 50, dtype=np.uint8), x[500:2500], rng.integers(0, 8, 70, dtype=np.uint8)]),   # one copy
        np.concatenate([x, x[100:2000], x[300:1500], np.zeros(3, np.uint8)]),                                              # three copies, zero tail
        np.concatenate([x[:1500], x[:1500]]),                                                                              # a square: the copy ends at n
        oracle_mod.gen_enwik_like(6000, 3, 1024),
        np.tile(oracle_mod.gen_uniform(97, 5) & 3, 30),                                                                    # periodic: every group is large
    ]
    decided = total = 0
    for T in texts:
        T = np.ascontiguousarray(T, dtype=np.uint8)
        sa = oracle_mod.divsufsort(T).astype(np.int64)
        isa = np.empty(T.size, np.int64)
        isa[sa] = np.arange(T.size)
        for h in (4, 16):
            for maxg in (2, 4):
                ans = pair_chain_model(T, sa, h, maxg)
                for (a, b), st in ans.items():
                    total += 1
                    if st != 3:
                        decided += 1
                        assert (st == 1) == (isa[a] < isa[b]), (h, maxg, a, b, st)
    assert decided > total // 2                        # (and it is worth it: most pairs of these inputs are decided)


# ---------------------------------------------------------------------------------------------------------------
# dq_runs.h: runs of one byte decided by their own structure.  A numpy model of the doubling rounds with the run
# lengths, the run-order round and the per-member offsets, validated against the oracle BEFORE any kernel runs.
def run_lengths(T, period=1):
    """RL[i] = how far the text goes on from i repeating itself with the given period: the largest L <= n - i with
    T[k] == T[k + period] for every k in [i, i + L - period).  period 1: the number of equal bytes from i on."""
    n = T.size
    ones = np.zeros(n + 1, np.int64)                                   # consecutive k >= i with T[k] == T[k + period]
    for i in range(n - period - 1, -1, -1):
        if T[i] == T[i + period]:
            ones[i] = ones[i + 1] + 1
    return np.minimum(ones[:n] + period, n - np.arange(n))


def doubling_with_runs(T, h0=2, period=1):
    """Prefix doubling from depth h0 (ranks = group starts of the h0-byte keys), with the run rules of dq_runs.h.
    period > 1: the same rules for stretches that repeat with that period (h0 >= period)."""
    assert h0 >= period
    n = T.size
    P = np.concatenate([T, np.zeros(h0, np.uint8)])
    cols = [P[b:b + n] for b in range(h0)]                             # the first h0 bytes of every suffix, zero padded
    # (zero padding can tie a suffix that ends with one that goes on with zero bytes: the valid length breaks it)
    ln = np.minimum(n - np.arange(n), h0)
    order = np.lexsort([ln] + cols[::-1])
    head = np.ones(n, bool)
    diff = ln[order][1:] != ln[order][:-1]
    for c in cols:
        diff |= c[order][1:] != c[order][:-1]
    head[1:] = diff
    rank_sorted = np.maximum.accumulate(np.where(head, np.arange(n), 0))
    ISA = np.empty(n, np.int64)
    ISA[order] = rank_sorted
    RL = run_lengths(T, period)
    Tx = np.concatenate([T.astype(np.int64), [-1]])                    # the end of the text sorts before every byte

    def rebucket(r, k2):
        o = np.lexsort((k2, r))
        return o, r[o], k2[o]

    h = h0
    s = np.arange(n)
    rounds = 0
    special = True
    while True:
        r = ISA[s]
        if special:                                                   # the run-order round: depth h stays
            e = s + RL[s]
            # the byte that ends the stretch against the byte the repetition would have put there (period 1: T[s])
            down = Tx[np.minimum(e, n)] < Tx[np.maximum(np.minimum(e, n) - period, 0)]
            enc = np.where(down, RL[s], (1 << 31) | ((1 << 31) - 1 - RL[s]))
            k2 = np.where(RL[s] >= h, enc, 0)
        else:
            off = np.where(RL[s] > h, RL[s], h)
            q = s + off
            k2 = np.where(q < n, ISA[np.minimum(q, n - 1)] + h, np.where(off > h, 0, n - 1 - s))
        o, rs, ks = rebucket(r, k2)
        ss = s[o]
        m = ss.size
        newhead = np.ones(m, bool)
        newhead[1:] = (rs[1:] != rs[:-1]) | (ks[1:] != ks[:-1])
        ghead = np.ones(m, bool)
        ghead[1:] = rs[1:] != rs[:-1]
        j = np.arange(m)
        nh = np.maximum.accumulate(np.where(newhead, j, -1))
        gh = np.maximum.accumulate(np.where(ghead, j, -1))
        ISA[ss] = rs + (nh - gh)
        rounds += 1
        if special:
            special = False
        else:
            h *= 2
        # stop when every rank is unique
        if np.unique(ISA).size == n:
            break
        assert h < 4 * n + 8, "no progress"
    SA = np.empty(n, np.int64)
    SA[ISA] = np.arange(n)
    return SA, rounds


def test_run_order_rule_and_offsets_give_the_suffix_array(oracle_mod):
    rng = np.random.default_rng(5)
    texts = [np.zeros(300, np.uint8), np.array([1] * 40 + [0] + [1] * 50 + [2] + [1] * 45, np.uint8)]
    for _ in range(40):
        parts = []
        for _ in range(int(rng.integers(2, 14))):
            if rng.random() < 0.6:
                parts.append(np.full(int(rng.integers(1, 120)), int(rng.choice([0, 1, 1, 2])), np.uint8))
            else:
                parts.append(rng.integers(0, 3, int(rng.integers(1, 6)), dtype=np.uint8))
        texts.append(np.concatenate(parts))
    for T in texts:
        for h0 in (1, 2, 4, 8):
            SA, rounds = doubling_with_runs(T, h0)
            assert np.array_equal(SA, oracle_mod.divsufsort(T).astype(np.int64)), (T.size, h0)
    # one long run needs a constant number of rounds, not log2(length)
    T = np.concatenate([np.zeros(5000, np.uint8), [3], np.zeros(4000, np.uint8), [1, 2]]).astype(np.uint8)
    SA, rounds = doubling_with_runs(T, 2)
    assert np.array_equal(SA, oracle_mod.divsufsort(T).astype(np.int64))
    assert rounds <= 4, rounds


def test_run_rules_hold_for_stretches_of_any_period(oracle_mod):
    """The generalisation the late run rounds use (dq_runs.h with a period): a stretch that repeats itself with period p
    is also one of period 2p, 4p ...; the rules with period P decide every stretch whose period divides P."""
    rng = np.random.default_rng(16)
    for trial in range(60):
        parts = []
        for _ in range(int(rng.integers(2, 12))):
            u = rng.random()
            if u < 0.55:                                               # a periodic stretch
                p = int(rng.choice([1, 2, 2, 4, 8, 3]))
                unit = rng.integers(0, 3, p, dtype=np.uint8)
                parts.append(np.resize(unit, int(rng.integers(1, 150))))
            elif u < 0.7 and parts:                                    # an earlier stretch again
                src = np.concatenate(parts)
                a = int(rng.integers(0, src.size))
                parts.append(src[a:a + int(rng.integers(1, 90))].copy())
            else:
                parts.append(rng.integers(0, 3, int(rng.integers(1, 6)), dtype=np.uint8))
        T = np.concatenate(parts).astype(np.uint8)
        ref = oracle_mod.divsufsort(T).astype(np.int64)
        for period, h0 in ((1, 1), (2, 2), (2, 4), (4, 4), (8, 8), (8, 16), (16, 16), (3, 4)):
            SA, _ = doubling_with_runs(T, h0, period)
            assert np.array_equal(SA, ref), (trial, T.size, period, h0)
    # a long stretch of period 2 takes a constant number of rounds with period 2 (or 8), log2(length) with period 1
    T = np.concatenate([np.resize(np.array([1, 2], np.uint8), 6001), [0], np.resize(np.array([1, 2], np.uint8), 3000), [3]]).astype(np.uint8)
    ref = oracle_mod.divsufsort(T).astype(np.int64)
    r = {}
    for period in (1, 2, 8):
        SA, r[period] = doubling_with_runs(T, 8, period)
        assert np.array_equal(SA, ref)
    assert r[2] <= 5 and r[8] <= 5 and r[1] > r[2] + 3, r


# ---------------------------------------------------------------------------------------------------------------
# dq_small_groups.h, twin_mark_kernel / twin_compact_kernel: doubled texts (bzip2's block + block).  Suffix i + half is a
# proper prefix of suffix i, so the pair stays tied for half - i characters although its order is known: the shorter one
# first.  Between the rounds every tie group that is exactly such a pair gets its two final ranks and leaves the list.
def doubling_without_twins(T2, h0=2):
    n = T2.size
    half = n // 2
    P = np.concatenate([T2, np.zeros(h0, np.uint8)])
    cols = [P[b:b + n] for b in range(h0)]
    ln = np.minimum(n - np.arange(n), h0)
    order = np.lexsort([ln] + cols[::-1])
    head = np.ones(n, bool)
    diff = ln[order][1:] != ln[order][:-1]
    for c in cols:
        diff |= c[order][1:] != c[order][:-1]
    head[1:] = diff
    ISA = np.empty(n, np.int64)
    ISA[order] = np.maximum.accumulate(np.where(head, np.arange(n), 0))
    h, rounds, written = h0, 0, 0
    while True:
        # the list: members of groups of more than one, in rank order
        cnt = np.bincount(ISA, minlength=n)
        s = np.flatnonzero(cnt[ISA] > 1)
        if s.size == 0:
            break
        s = s[np.argsort(ISA[s], kind="stable")]
        r = ISA[s]
        # ---- the twin step: groups of exactly two members that lie `half` apart are written down
        gsize = cnt[r]
        first = np.ones(s.size, bool)
        first[1:] = r[1:] != r[:-1]
        pair_head = np.flatnonzero(first & (gsize == 2))
        a, b = s[pair_head], s[pair_head + 1]
        twin = np.abs(a - b) == half
        hi, lo = np.maximum(a, b)[twin], np.minimum(a, b)[twin]
        ISA[hi] = r[pair_head][twin]                                   # the shorter suffix first
        ISA[lo] = r[pair_head][twin] + 1
        written += int(twin.sum())
        keep = np.ones(s.size, bool)
        keep[pair_head[twin]] = False
        keep[pair_head[twin] + 1] = False
        s, r = s[keep], r[keep]
        if s.size == 0:
            break
        # ---- one plain doubling round on what is left (ranks finer than depth h are welcome as keys)
        q = s + h
        k2 = np.where(q < n, ISA[np.minimum(q, n - 1)] + h, n - 1 - s)
        o = np.lexsort((k2, r))
        ss, rs, ks = s[o], r[o], k2[o]
        m = ss.size
        newhead = np.ones(m, bool)
        newhead[1:] = (rs[1:] != rs[:-1]) | (ks[1:] != ks[:-1])
        ghead = np.ones(m, bool)
        ghead[1:] = rs[1:] != rs[:-1]
        j = np.arange(m)
        ISA[ss] = rs + (np.maximum.accumulate(np.where(newhead, j, -1)) - np.maximum.accumulate(np.where(ghead, j, -1)))
        h *= 2
        rounds += 1
        assert h < 4 * n + 8, "no progress"
    SA = np.empty(n, np.int64)
    SA[ISA] = np.arange(n)
    return SA, rounds, written


def test_twin_pairs_of_a_doubled_text_leave_the_list_early(oracle_mod):
    rng = np.random.default_rng(23)
    blocks = [rng.integers(0, 256, 3000, dtype=np.uint8), rng.integers(0, 2, 700, dtype=np.uint8), np.zeros(64, np.uint8),
              np.resize(np.array([0, 0, 0, 0, 251], np.uint8), 1200), np.resize(np.array([7, 9], np.uint8), 300),
              np.frombuffer(b"abracadabra" * 30 + b"x", np.uint8), np.array([5], np.uint8), np.array([1, 1], np.uint8)]
    for _ in range(25):
        parts = [np.resize(rng.integers(0, 3, int(rng.integers(1, 6)), dtype=np.uint8), int(rng.integers(1, 200))) for _ in range(int(rng.integers(1, 9)))]
        blocks.append(np.concatenate(parts).astype(np.uint8))
    for B in blocks:
        T2 = np.concatenate([B, B]).astype(np.uint8)
        ref = oracle_mod.divsufsort(T2).astype(np.int64)
        for h0 in (1, 2, 8):
            SA, rounds, written = doubling_without_twins(T2, h0)
            assert np.array_equal(SA, ref), (B.size, h0)
    # a block of random bytes: its real ties end with the first round or two, its 3000 pairs would have lasted eleven
    SA, rounds, written = doubling_without_twins(np.concatenate([blocks[0], blocks[0]]), 2)
    assert rounds <= 2 and written >= 2800, (rounds, written)


def test_run_length_passes_model():
    """The three-pass run-length computation of dq_runs.h (segments of 16, chunks of 4096, carry across chunks) in numpy."""
    rng = np.random.default_rng(9)

    def three_pass(T, seg=16, chunk=64):
        n = T.size
        nch = (n + chunk - 1) // chunk
        lead = np.zeros(nch, np.int64)
        link = np.zeros(nch, bool)
        local = np.zeros(n, np.int64)
        for c in range(nch):
            lo, hi = c * chunk, min((c + 1) * chunk, n)
            for i in range(hi - 1, lo - 1, -1):
                local[i] = local[i + 1] + 1 if i + 1 < hi and T[i + 1] == T[i] else 1
            lead[c] = local[lo]
            link[c] = lo + lead[c] == hi and hi < n and T[hi] == T[lo]
        carry = np.zeros(nch + 1, np.int64)
        for c in range(nch - 1, -1, -1):
            carry[c] = lead[c] + (carry[c + 1] if link[c] else 0)
        RL = local.copy()
        for c in range(nch):
            lo, hi = c * chunk, min((c + 1) * chunk, n)
            for i in range(lo, hi):
                if i + local[i] == hi and hi < n and T[hi] == T[i]:
                    RL[i] += carry[c + 1]
        return RL

    for _ in range(30):
        parts = [np.full(int(rng.integers(1, 200)), int(rng.integers(0, 2)), np.uint8) for _ in range(int(rng.integers(1, 9)))]
        T = np.concatenate(parts)
        assert np.array_equal(three_pass(T), run_lengths(T))


# ---------------------------------------------------------------------------------------------------------------
# dq_mid_groups.h: which workgroup finishes / forwards which list entry.  A numpy model of the kernel's ownership
# rules (left halo of ranks, right overhang, group closed iff its closer is seen) run over random group layouts:
# every list entry must be claimed by exactly one tile, as a member of a group finished in LDS or as a member of a
# large group handed to the radix list -- and both sides of a tile boundary must agree on which.
def mid_kernel_claims(ranks, span=64, G=16):
    m = ranks.size
    tile = span - G
    scan = G + span
    mid = np.zeros(m, np.int64)          # times claimed as a member of a group finished in LDS
    large = np.zeros(m, np.int64)        # times forwarded to the radix list
    NONE = -1
    for b in range((m + tile - 1) // tile):
        j0 = b * tile
        r = np.array([ranks[j] if 0 <= j < m else NONE for j in range(j0 - G, j0 - G + scan + 1)])   # [scan] = the closer
        head = np.zeros(scan, bool)
        for c in range(scan):
            head[c] = (j0 - G <= 0) if c == 0 else r[c] != r[c - 1]
        hp = np.full(scan, -1)
        run = -1
        for c in range(scan):
            run = c if head[c] else run
            hp[c] = run
        gsize = {}
        for c in range(1, scan):
            if head[c] and hp[c - 1] >= 0:
                gsize[hp[c - 1]] = c - hp[c - 1]
        if hp[scan - 1] >= 0:
            gsize[hp[scan - 1]] = (scan - hp[scan - 1]) if r[scan] != r[scan - 1] else None      # None: open
        for e in range(span):
            j = j0 + e
            if not (0 <= j < m):
                continue
            hc = hp[G + e]
            gs = gsize.get(hc) if hc >= 0 else None
            is_large = hc < 0 or gs is None or gs > G
            ghead = hc - G
            if not is_large and 0 <= ghead < tile:
                mid[j] += 1
            if is_large and e < tile:
                large[j] += 1
    return mid, large


def test_mid_group_tiles_claim_every_entry_exactly_once():
    rng = np.random.default_rng(21)
    for trial in range(60):
        sizes = []
        total = int(rng.integers(1, 700))
        while sum(sizes) < total:
            u = rng.random()
            sizes.append(int(rng.integers(2, 6)) if u < 0.5 else int(rng.integers(6, 17)) if u < 0.8
                         else int(rng.integers(17, 40)) if u < 0.95 else int(rng.integers(40, 300)))
        ranks = np.concatenate([np.full(s, 1000 * i) for i, s in enumerate(sizes)])
        mid, large = mid_kernel_claims(ranks)
        assert np.all(mid + large == 1), (trial, sizes)
        # a group of <= G members is finished in LDS, a longer one goes to the radix list -- whole groups either way
        pos = 0
        for s in sizes:
            if s <= 16:
                assert np.all(mid[pos:pos + s] == 1), (trial, pos, s)
            else:
                assert np.all(large[pos:pos + s] == 1), (trial, pos, s)
            pos += s


# ---------------------------------------------------------------------------------------------------------------
# dq_match_search.h, second stage of the window kernel: the mailbox word  ticket << 44 | (4095 - position) << 32 | length
# kept by atomic max and never reset.  Model: whatever order the first-stage waves of several launches publish in,
# the word a second stage reads is (a) of its own launch iff any wave of that launch published, with (b) the SMALLEST
# position of that launch; an older launch's word never passes the ticket test; lengths never disturb the order.
def mail_word(ticket, pos, length):
    assert 0 <= pos < 4096 and 0 <= length < (1 << 32) and 0 < ticket < (1 << 20)
    return (ticket << 44) | ((4095 - pos) << 32) | length


def test_mailbox_word_orders_by_ticket_then_smallest_position():
    rng = np.random.default_rng(8)
    mail = 0
    for ticket in range(1, 200):
        k = int(rng.integers(0, 6))                                   # long matches found by this launch's waves
        pubs = [(int(rng.integers(0, 4096)), int(rng.integers(32, 1 << 31))) for _ in range(k)]
        order = rng.permutation(k)
        for i in order:                                               # publication order is arbitrary
            mail = max(mail, mail_word(ticket, *pubs[i]))
        won = (mail >> 44) == ticket                                  # the second stage's test
        assert won == (k > 0)
        if won:
            pos = 4095 - ((mail >> 32) & 0xfff)
            length = mail & 0xffffffff
            best = min(p for p, _ in pubs)
            assert pos == best and length in [l for p, l in pubs if p == best]
    # the host stops using the second stage before the ticket field would overflow (ticket < 2^20 - 2)
    assert mail_word((1 << 20) - 3, 0, (1 << 32) - 1) < (1 << 64)


# ---- the device's anchor search (dq_anchor_scan.h): windowed evaluation of Diff.cs:100-125 ----
def _edited_pairs(oracle_mod, rng, trials):
    for trial in range(trials):
        n = int(rng.integers(1, 3000))
        kind = trial % 4
        old = oracle_mod.gen_enwik_like(n, 900 + trial, 512) if kind == 1 else oracle_mod.gen_uniform(n, 900 + trial)
        if kind == 2:
            old = np.tile(old[: max(1, n // 23)], 23)[:n].copy()               # periodic: alignments that carry on
        if kind == 3:
            old = (old & 1).astype(np.uint8)                                    # two symbols: long matches everywhere
        new = bytearray(old.tobytes())
        for _ in range(int(rng.integers(0, 8))):
            a, ln, k = int(rng.integers(0, max(1, len(new)))), int(rng.integers(1, 120)), int(rng.integers(0, 4))
            if k == 0:
                new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
            elif k == 1:
                del new[a:a + ln]
            elif k == 2:
                new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
            else:
                new[int(rng.integers(0, max(1, len(new)))):0] = new[a:a + ln]
        if trial % 17 == 5:
            new = bytearray(oracle_mod.gen_uniform(int(rng.integers(0, 400)), trial).tobytes())   # unrelated
        yield trial, old, np.frombuffer(bytes(new), dtype=np.uint8)


def test_windowed_anchor_search_equals_the_loop(oracle_mod):
    import ctypes
    import os
    import subprocess
    import anchor_model
    from conftest import ROOT
    native = os.path.join(ROOT, "tests", "native")
    so, src = os.path.join(native, "libscan_harness.so"), os.path.join(native, "scan_harness.cpp")
    hdr = os.path.join(ROOT, "deltaq_amd", "csrc", "dq_bsdiff.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.t_scan_from_anchors.restype = ctypes.c_int64
    L.t_scan_from_anchors.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_int64] + [ctypes.c_void_p] * 6
    rng = np.random.default_rng(2024)
    stops = windows = bound_stops = 0
    for trial, old, new in _edited_pairs(oracle_mod, rng, 120):
        sa = oracle_mod.divsufsort(old)
        pos, ln = oracle_mod.bsdiff_search(old, sa, new) if new.size else (np.zeros(0, np.int32), np.zeros(0, np.int32))
        O, N = old.tolist(), new.tolist()
        want = anchor_model.literal_anchors(O, N, pos, ln)
        cap = int(rng.choice([0, 4, 16, 64]))
        noise = set(rng.integers(0, max(1, new.size), new.size // 7).tolist())
        capped = (lambda j: ln[j] > cap or j in noise) if cap else (lambda j: False)
        for (w1, w2, ex) in ((8, 32, 16), (1, 1, 0), (128, 512, 64), (3, 5, 2)):
            got, st = anchor_model.windowed_anchors(O, N, pos, ln, capped, w1, w2, ex)
            assert got == want, (trial, w1, w2, ex, cap)
            stops += st["stops"]; windows += st["windows"]
            # counts of matches under another alignment as upper bounds (the kernel counts the front of a long match only)
            slack = rng.integers(0, 40, new.size + 1) * (rng.integers(0, 3, new.size + 1) > 0)
            got, st = anchor_model.windowed_anchors(O, N, pos, ln, capped, w1, w2, ex, loose=lambda j: slack[j])
            assert got == want, (trial, w1, w2, ex, cap, "bounds")
            bound_stops += st["stops"]
        # the product's emitter on those anchors = the oracle's streams
        pairs = np.array(want, dtype=np.int64).reshape(-1)
        m = new.size
        ctrl = np.empty(24 * (m + 2), np.uint8); diff = np.empty(max(m, 1), np.uint8); extra = np.empty(max(m, 1), np.uint8)
        lens = np.zeros(3, np.int64)
        oc, nc = np.ascontiguousarray(old), np.ascontiguousarray(new)
        L.t_scan_from_anchors(oc.ctypes.data, oc.size, nc.ctypes.data if m else None, m, pairs.ctypes.data if pairs.size else None,
                              pairs.size // 2, ctrl.ctypes.data, lens.ctypes.data, diff.ctypes.data, lens.ctypes.data + 8,
                              extra.ctypes.data, lens.ctypes.data + 16)
        wc, wd, we, _ = oracle_mod.bsdiff_scan(old, sa, new)
        raw = ctrl[:lens[0]].reshape(-1, 8).astype(np.int64)
        mag = sum((raw[:, i] & (0x7f if i == 7 else 0xff)) << (8 * i) for i in range(8))
        trip = np.where(raw[:, 7] & 0x80, -mag, mag).reshape(-1, 3)
        assert np.array_equal(trip, wc), trial
        assert np.array_equal(diff[:lens[1]], wd) and np.array_equal(extra[:lens[2]], we), trial
    assert stops > 100 and windows > 1000            # the stop-point path was exercised
    assert bound_stops > stops                       # ... and so were the bounds that could not decide


# ---- dq_tail.h: the last doubling rounds in one workgroup; dq_mid_groups.h's shifted radix-list keys -----------------
def tail_rounds_model(T, h0, steps=1):
    """tail_rounds_kernel on the CPU: start from the groups of suffixes that share their first h0 bytes (rank = the SA
    index of the group's first member), keep only the tied ones as the list, then round after round: key2 = ISA[s + h] + h
    (n - 1 - s past the end), place = #(smaller key2) + #(equal before), SA / ISA written for what is decided, the rest
    compacted in slot order.  Returns (SA, rounds)."""
    n = T.size
    pad = np.concatenate([T, np.zeros(h0, np.uint8)]).astype(np.int64)
    keys = [tuple(pad[i:i + h0]) + (min(h0, n - i),) for i in range(n)]       # (bytes, valid length): a proper prefix first
    order = sorted(range(n), key=lambda i: keys[i])
    SA = np.full(n, -1, np.int64)
    ISA = np.zeros(n, np.int64)
    lst = []                                                                  # (rank, suffix), groups adjacent
    p = 0
    while p < n:
        q = p
        while q < n and keys[order[q]] == keys[order[p]]:
            q += 1
        for j in range(p, q):
            ISA[order[j]] = p
        if q - p == 1:
            SA[p] = order[p]
        else:
            lst += [(p, order[j]) for j in range(p, q)]
        p = q
    h, rounds = h0, 0
    while lst:
        rounds += 1
        assert rounds < 80
        # steps > 1: the key is the tuple of the ranks h, 2h, ... steps*h bytes further on; a suffix that ends inside stretch
        # j has (n - 1 - s) - j*h there (< h: below every in-range value, the shorter suffix first) and 0 behind it
        def key_of(s):
            out = []
            for j in range(steps):
                q, qp = s + (j + 1) * h, s + j * h
                out.append(int(ISA[q]) + h if q < n else (n - 1 - s - j * h if qp < n else 0))
            return tuple(out) if steps > 1 else out[0]
        key2 = [key_of(s) for _, s in lst]
        nxt, slots, i = [], [None] * len(lst), 0
        while i < len(lst):
            j = i
            while j < len(lst) and lst[j][0] == lst[i][0]:
                j += 1
            for a in range(i, j):                                             # place inside the group
                less = sum(1 for b in range(i, j) if key2[b] < key2[a])
                eq = sum(1 for b in range(i, j) if key2[b] == key2[a])
                eq_before = sum(1 for b in range(i, a) if key2[b] == key2[a])
                slots[i + less + eq_before] = (lst[a][0] + less, lst[a][1], eq > 1, less != 0)
            i = j
        for r, s, tied, moved in slots:
            if moved:
                ISA[s] = r
            if tied:
                nxt.append((r, s))
            else:
                SA[r] = s
        lst = nxt
        h *= steps + 1
    return SA, rounds


def test_tail_rounds_model_gives_the_suffix_array(oracle_mod):
    rng = np.random.default_rng(41)
    cases = [np.zeros(300, np.uint8), np.frombuffer(b"abracadabra" * 30, np.uint8), rng.integers(0, 2, 500).astype(np.uint8),
             np.concatenate([rng.integers(0, 256, 400), np.zeros(9)]).astype(np.uint8)]
    x = rng.integers(0, 4, 300).astype(np.uint8)
    cases.append(np.concatenate([x, x[50:250], x[:100]]))                      # long repeats: many rounds
    for T in cases:
        T = np.ascontiguousarray(T)
        for h0 in (1, 2, 4):
            SA, rounds = tail_rounds_model(T, h0)
            assert np.array_equal(SA, oracle_mod.divsufsort(T).astype(np.int64)), (T.size, h0)
            assert rounds >= 1
            # three ranks a member and round (the chained rounds of short lists, the tail kernel): the same array in
            # fewer rounds
            SA3, rounds3 = tail_rounds_model(T, h0, steps=3)
            assert np.array_equal(SA3, SA), (T.size, h0, "three steps")
            assert rounds3 <= rounds


def test_shifted_rank_field_keeps_large_groups_apart_in_order():
    """The radix list of an LDS-class round holds groups of MORE than cap members, so the ranks (first SA index) of two
    of them differ by more than cap = 2^shift: rank >> shift is still strictly increasing over the groups, and
    (rank >> shift) << (kbits + shift) | key2 << shift | (rank & (2^shift - 1)) sorts like rank << kbits | key2 when only
    the bits above `shift` are looked at -- while the full rank can be put together again from both ends of the word."""
    rng = np.random.default_rng(7)
    for shift in (8, 9, 10):
        cap = 1 << shift
        sizes = rng.integers(cap + 1, 4 * cap, 200)
        gaps = rng.integers(0, 3 * cap, 200)                                  # resolved suffixes / small groups in between
        ranks, r = [], int(rng.integers(0, 1000))
        for g, gap in zip(sizes, gaps):
            ranks.append(r)
            r += int(g) + int(gap)
        kbits = 30
        shifted = [x >> shift for x in ranks]
        assert all(a < b for a, b in zip(shifted, shifted[1:]))
        for x in ranks[:50]:
            for k2 in (0, 1, (1 << kbits) - 1):
                word = ((x >> shift) << (kbits + shift)) | (k2 << shift) | (x & (cap - 1))
                assert word < (1 << 64)
                field = word >> shift
                assert field == ((x >> shift) << kbits) | k2                                  # what the digit passes sort by
                assert ((field >> kbits) << shift) | (word & (cap - 1)) == x                  # what the rebucket pass reads back
