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


# ---- the PRODUCT's scan loop (deltaq_amd/csrc/dq_bsdiff.h) on the CPU: Search answers come from a table the oracle
#      filled, the loop itself is the shipped code (tests/native/scan_harness.cpp) ----
@pytest.fixture(scope="module")
def scan_harness():
    import ctypes
    import os
    import subprocess
    from conftest import ROOT
    native = os.path.join(ROOT, "tests", "native")
    so, src = os.path.join(native, "libscan_harness.so"), os.path.join(native, "scan_harness.cpp")
    hdr = os.path.join(ROOT, "deltaq_amd", "csrc", "dq_bsdiff.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.t_scan_loop.restype = ctypes.c_int64
    L.t_scan_loop.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 8
    L.t_packed_roundtrip.restype = ctypes.c_int64
    L.t_packed_roundtrip.argtypes = [ctypes.c_int64]

    def run(old, new, pos, ln):
        m = new.size
        ctrl = np.empty(24 * (m + 1), np.uint8)
        diff = np.empty(max(m, 1), np.uint8)
        extra = np.empty(max(m, 1), np.uint8)
        lens = np.zeros(3, np.int64)
        pos = np.ascontiguousarray(pos, np.int64)
        ln = np.ascontiguousarray(ln, np.int64)
        old = np.ascontiguousarray(old)
        new = np.ascontiguousarray(new)
        r = L.t_scan_loop(old.ctypes.data, old.size, new.ctypes.data, m, pos.ctypes.data, ln.ctypes.data,
                          ctrl.ctypes.data, lens.ctypes.data, diff.ctypes.data, lens.ctypes.data + 8,
                          extra.ctypes.data, lens.ctypes.data + 16)
        assert r >= 0
        raw = ctrl[:lens[0]].reshape(-1, 8).astype(np.int64)
        mag = sum((raw[:, i] & (0x7f if i == 7 else 0xff)) << (8 * i) for i in range(8))
        trip = np.where(raw[:, 7] & 0x80, -mag, mag).reshape(-1, 3)
        return trip, diff[:lens[1]].copy(), extra[:lens[2]].copy(), int(r)
    run.lib = L
    return run


def test_product_scan_loop_makes_the_references_decisions(oracle_mod, scan_harness):
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(1, 12000))
        old = oracle_mod.gen_enwik_like(n, 300 + trial, 1024) if trial % 2 else oracle_mod.gen_uniform(n, 300 + trial)
        if trial % 5 == 0:
            old = np.tile(old[: max(1, n // 37)], 37)[:n].copy()            # periodic: long carried alignments
        new = bytearray(old.tobytes())
        for _ in range(int(rng.integers(0, 10))):
            a = int(rng.integers(0, max(1, len(new))))
            ln = int(rng.integers(1, 300))
            k = int(rng.integers(0, 4))
            if k == 0:
                new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
            elif k == 1:
                del new[a:a + ln]
            elif k == 2:
                new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
            else:
                new[int(rng.integers(0, max(1, len(new)))):0] = new[a:a + ln]
        if trial == 7:
            new = bytearray()
        new = np.frombuffer(bytes(new), dtype=np.uint8)
        sa = oracle_mod.divsufsort(old)
        pos, ln = oracle_mod.bsdiff_search(old, sa, new) if new.size else (np.zeros(0, np.int32), np.zeros(0, np.int32))
        want = oracle_mod.bsdiff_scan(old, sa, new)
        got = scan_harness(old, new, pos, ln)
        assert np.array_equal(got[0], want[0]), trial
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), trial
        assert got[3] == want[3], trial


def test_product_scan_loop_on_stretches_the_files_do_not_share(oracle_mod, scan_harness):
    """The extensions step over eight positions at once where none of them can be a new best (dq_bsdiff.h): unrelated
    files and long inserted stretches, over alphabets of 2, 4 and 256 symbols -- with few symbols every second or fourth
    byte agrees by chance, so the bound the step rests on is met from both sides."""
    rng = np.random.default_rng(11)
    for trial in range(36):
        n = int(rng.integers(200, 20000))
        mask = (1, 3, 255)[trial % 3]
        old = oracle_mod.gen_uniform(n, 700 + trial) & mask
        kind = trial % 4
        if kind == 0:
            new = oracle_mod.gen_uniform(int(rng.integers(100, 20000)), 900 + trial) & mask          # nothing in common
        else:
            x = bytearray(old.tobytes())
            for _ in range(kind):                                                                     # long foreign stretches
                a = int(rng.integers(0, len(x)))
                x[a:a] = (oracle_mod.gen_uniform(int(rng.integers(500, 6000)), 1100 + trial) & mask).tobytes()
            new = np.frombuffer(bytes(x), dtype=np.uint8)
        new = np.ascontiguousarray(new, np.uint8)
        sa = oracle_mod.divsufsort(old)
        pos, ln = oracle_mod.bsdiff_search(old, sa, new)
        want = oracle_mod.bsdiff_scan(old, sa, new)
        got = scan_harness(old, new, pos, ln)
        assert np.array_equal(got[0], want[0]), (trial, n, new.size)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), trial
        assert got[3] == want[3], trial


def test_product_packed_longs(scan_harness):
    for v in (0, 1, -1, 127, 128, -128, 255, 256, 2**31 - 1, -2**31, 2**62, -(2**62), 2**63 - 1, -(2**63 - 1)):
        assert scan_harness.lib.t_packed_roundtrip(v) == v
