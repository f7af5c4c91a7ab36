"""CPU tests of the oracle's restatement of the suffix array's consumer (oracle/bsdiff_scan.c):
Diff.Create's Search and scan loop (Diff.cs:91-298) and Patch.ApplyInternal (Patch.cs:95-168) on raw streams."""
import numpy as np
import pytest


def brute_search(old, new, scan):
    """Search's specification evaluated naively: g = number of suffixes of old below the query;
    candidates I[max(g-1, 0)] and its upper neighbour (I[n] = 0); longer match wins, ties to the upper one."""
    O, q, n = bytes(old), bytes(new[scan:]), len(old)
    sa = sorted(range(n), key=lambda i: O[i:])
    I = sa + [0]
    g = sum(1 for i in sa if O[i:] < q)
    start = max(g - 1, 0)
    end = start + 1 if n > 0 else 0

    def ml(a, b):
        k = 0
        while k < len(a) and k < len(b) and a[k] == b[k]:
            k += 1
        return k
    x, y = ml(O[I[start]:], q), ml(O[I[end]:], q)
    return (I[start], x) if x > y else (I[end], y)


def test_search_restatement_against_its_specification(oracle_mod):
    rng = np.random.default_rng(3)
    for trial in range(150):
        n, m = int(rng.integers(0, 60)), int(rng.integers(1, 60))
        sigma = int(rng.choice([2, 3, 256]))
        old = rng.integers(0, sigma, n, dtype=np.uint8)
        new = rng.integers(0, sigma, m, dtype=np.uint8)
        if trial % 4 == 0 and n > 10:
            k = min(m, n - 3)
            new[:k] = old[3:3 + k]
        sa = oracle_mod.divsufsort(old)
        pos, ln = oracle_mod.bsdiff_search(old, sa, new)
        for s in range(m):
            assert (int(pos[s]), int(ln[s])) == brute_search(old, new, s), (trial, s)
        p64, l64 = oracle_mod.bsdiff_search(old, sa.astype(np.int64), new)
        assert np.array_equal(p64, pos) and np.array_equal(l64, ln)


@pytest.mark.parametrize("size", [0, 1, 512, 999, 1024, 4096])
def test_roundtrip_like_BsDiffTests(oracle_mod, size):
    # BsDiffTests.cs:30-53: GetBuffer(size) twice (same seed: identical buffers) and the explicit "_Identical" case
    old = oracle_mod.net_random_bytes(size)
    new = oracle_mod.net_random_bytes(size)
    sa = oracle_mod.divsufsort(old)
    ctrl, diff, extra, _ = oracle_mod.bsdiff_scan(old, sa, new)
    assert np.array_equal(oracle_mod.bspatch_apply(old, ctrl, diff, extra, new.size), new)
    if size:
        assert ctrl[:, :2].tolist() == [[size, 0]] and not diff.any() and extra.size == 0


def test_roundtrip_on_edited_buffers(oracle_mod):
    rng = np.random.default_rng(11)
    for trial in range(30):
        n = int(rng.integers(1, 20000))
        old = oracle_mod.gen_enwik_like(n, 100 + trial, 2048) if trial % 2 else oracle_mod.gen_uniform(n, 100 + trial)
        new = bytearray(old.tobytes())
        for _ in range(int(rng.integers(0, 12))):                 # inserts, deletions, overwrites, moved blocks
            k = int(rng.integers(0, 4))
            a = int(rng.integers(0, max(1, len(new))))
            ln = int(rng.integers(1, 200))
            if k == 0:
                new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
            elif k == 1:
                del new[a:a + ln]
            elif k == 2:
                new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
            else:
                blk = new[a:a + ln]
                b = int(rng.integers(0, max(1, len(new))))
                new[b:b] = blk
        new = np.frombuffer(bytes(new), dtype=np.uint8)
        sa = oracle_mod.divsufsort(old)
        ctrl, diff, extra, searches = oracle_mod.bsdiff_scan(old, sa, new)
        assert np.array_equal(oracle_mod.bspatch_apply(old, ctrl, diff, extra, new.size), new), trial
        assert int(ctrl[:, 0].sum() + ctrl[:, 1].sum()) == new.size
        c64 = oracle_mod.bsdiff_scan(old, sa.astype(np.int64), new)
        assert np.array_equal(c64[0], ctrl) and np.array_equal(c64[1], diff) and np.array_equal(c64[2], extra)


def test_corrupt_patch_is_rejected(oracle_mod):
    old = oracle_mod.net_random_bytes(100)
    with pytest.raises(RuntimeError, match="Corrupt patch"):
        oracle_mod.bspatch_apply(old, np.array([[200, 0, 0]]), np.zeros(200, np.uint8), np.zeros(0, np.uint8), 100)
