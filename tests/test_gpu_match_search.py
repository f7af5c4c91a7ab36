"""GPU parity tests of the match search (dq_bsdiff_search_*, dq_match_search.h) against the oracle's restatement
of Diff.cs Search: every (pos, len) bit for bit, through the C ABI (host entry points via ctypes / numpy, device
entry points via torch tensors holding the suffix array the HIP sorter left on the device)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ms(backend_lib):
    from deltaq_amd import HipMatchSearch
    assert backend_lib.dq_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return HipMatchSearch(0)


@pytest.fixture(scope="module")
def ldss(backend_lib):
    from deltaq_amd import HipSuffixSort
    return HipSuffixSort(0)


def edited(rng, old, edits):
    new = bytearray(old.tobytes())
    for _ in range(edits):
        k = int(rng.integers(0, 4))
        a = int(rng.integers(0, max(1, len(new))))
        ln = int(rng.integers(1, 400))
        if k == 0:
            new[a:a] = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        elif k == 1:
            del new[a:a + ln]
        elif k == 2:
            new[a:a + ln] = rng.integers(0, 256, min(ln, len(new) - a), dtype=np.uint8).tobytes()
        else:
            new[a:a] = new[max(0, a - 3 * ln):max(0, a - 2 * ln)]
    return np.frombuffer(bytes(new), dtype=np.uint8)


def test_small_buffers_every_position(ms, ldss, oracle_mod):
    """Sizes of the reference's own tests (BsDiffTests.cs: 0, 1, 512, 999, 1024, 4096) and tiny alphabets,
    where equal candidates and the sentinel slot I[n] = 0 decide; every scan position, both index widths."""
    rng = np.random.default_rng(1)
    cases = []
    for size in (0, 1, 2, 3, 512, 999, 1024, 4096):
        old = oracle_mod.net_random_bytes(size)
        cases.append((old, old.copy()))
        cases.append((old, edited(rng, old, 6) if size else np.array([7, 7], np.uint8)))
    for sigma in (1, 2, 3):
        for n in (1, 5, 64, 300):
            cases.append((rng.integers(0, sigma, n, dtype=np.uint8), rng.integers(0, sigma, 2 * n + 1, dtype=np.uint8)))
    for old, new in cases:
        sa = oracle_mod.divsufsort(old)
        want = oracle_mod.bsdiff_search(old, sa, new, scan0=0, count=new.size + 1)       # scan == m too (empty query)
        got = ms.Search(sa, old, new, scan0=0, count=new.size + 1)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (old.size, new.size)
        got64 = ms.Search(sa.astype(np.int64), old, new, scan0=0, count=new.size + 1)
        assert np.array_equal(got64[0], want[0]) and np.array_equal(got64[1], want[1])


def test_differing_region_one_million_positions(ms, ldss, oracle_mod):
    """The regime the scan loop spends its Search calls in: new data that mostly differs from old (here: two
    independent random buffers with a few shared blocks).  16 MiB old, 10^6 consecutive scan positions; the
    suffix array comes from the HIP sorter and stays on the device."""
    import torch
    n = 16 << 20
    old = oracle_mod.gen_uniform(n, 0x5EED0500)
    new = oracle_mod.gen_uniform(n, 0x5EED0501)
    new[100_000:100_000 + 70_000] = old[5_000_000:5_070_000]            # shared blocks: long matches inside the range
    new[600_000:600_300] = old[123_456:123_756]
    d_old = torch.from_numpy(old).cuda()
    d_new = torch.from_numpy(new).cuda()
    d_sa = ldss.Sort(d_old)
    count = 1_000_000
    pos, ln = ms.Search(d_sa, d_old, d_new, scan0=0, count=count)
    assert pos.is_cuda and pos.dtype == torch.int32
    sa = d_sa.cpu().numpy()
    wpos, wlen = oracle_mod.bsdiff_search(old, sa, new, scan0=0, count=count)
    assert np.array_equal(ln.cpu().numpy(), wlen)
    assert np.array_equal(pos.cpu().numpy(), wpos)
    # an explicit position list, and the cap: positions inside the 70 000-byte block give up, the rest is exact
    scans = np.array([0, 99_999, 100_000, 100_001, 169_999, 170_000, 600_000, 999_999, n - 1, n], dtype=np.int64)
    p2, l2 = ms.Search(d_sa, d_old, d_new, scans=scans)
    w2 = oracle_mod.bsdiff_search(old, sa, new, scans=scans)
    assert np.array_equal(p2.cpu().numpy(), w2[0]) and np.array_equal(l2.cpu().numpy(), w2[1])
    p3, l3 = ms.Search(d_sa, d_old, d_new, scan0=99_000, count=4000, cap=1024)
    l3 = l3.cpu().numpy()
    w3 = oracle_mod.bsdiff_search(old, sa, new, scan0=99_000, count=4000)
    gave_up = l3 < 0
    assert gave_up.any() and np.all(w3[1][gave_up] >= 1024)
    assert np.array_equal(l3[~gave_up], w3[1][~gave_up]) and np.array_equal(p3.cpu().numpy()[~gave_up], w3[0][~gave_up])


def test_text_like_and_long_matches(ms, oracle_mod):
    """Text-like data (many near-equal candidates) and nearly identical files (match lengths of hundreds of
    kilobytes: the wave-cooperative comparison), sampled positions, host entry points."""
    rng = np.random.default_rng(5)
    old = oracle_mod.gen_enwik_like(3_000_000, 7, 64 * 1024)
    new = edited(rng, old, 40)
    sa = oracle_mod.divsufsort(old)
    scans = np.unique(np.concatenate([rng.integers(0, new.size, 20_000), np.arange(0, 3000), [new.size - 1, new.size]])).astype(np.int64)
    got = ms.Search(sa, old, new, scans=scans)
    want = oracle_mod.bsdiff_search(old, sa, new, scans=scans)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    assert int(want[1].max()) > 50_000                                  # long matches were exercised
    zeros = np.zeros(200_000, np.uint8)                                 # one letter: every suffix a prefix of the next
    sa0 = oracle_mod.divsufsort(zeros)
    q = np.zeros(150_000, np.uint8)
    got = ms.Search(sa0, zeros, q, scan0=0, count=300)
    want = oracle_mod.bsdiff_search(zeros, sa0, q, scan0=0, count=300)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])


def test_scan_loop_driven_by_device_search(ms, oracle_mod):
    """Diff.Create's scan loop (restated in the oracle) consumes Search only through (pos, len): feeding it the
    device's answers position by position must reproduce the oracle's control triples exactly."""
    rng = np.random.default_rng(9)
    old = oracle_mod.gen_enwik_like(200_000, 3, 8192)
    new = edited(rng, old, 25)
    sa = oracle_mod.divsufsort(old)
    pos, ln = ms.Search(sa, old, new, scan0=0, count=new.size)
    want = oracle_mod.bsdiff_search(old, sa, new, scan0=0, count=new.size)
    assert np.array_equal(pos, want[0]) and np.array_equal(ln, want[1])
    ctrl, diff, extra, _ = oracle_mod.bsdiff_scan(old, sa, new)
    assert np.array_equal(oracle_mod.bspatch_apply(old, ctrl, diff, extra, new.size), new)


def test_argument_checks(backend_lib):
    from deltaq_amd import _abi
    a = np.zeros(8, np.uint8)
    sa = np.arange(8, dtype=np.int32)
    out = np.zeros(4, np.int32)
    f = backend_lib.dq_bsdiff_search_i32
    assert f(a.ctypes.data, 8, sa.ctypes.data, a.ctypes.data, 8, None, 6, 4, 0, out.ctypes.data, out.ctypes.data, 0) == _abi.DQ_ERR_BAD_ARGS
    assert f(None, 8, sa.ctypes.data, a.ctypes.data, 8, None, 0, 4, 0, out.ctypes.data, out.ctypes.data, 0) == _abi.DQ_ERR_BAD_ARGS
    assert f(a.ctypes.data, 8, sa.ctypes.data, a.ctypes.data, 8, None, 0, 0, 0, None, None, 0) == _abi.DQ_OK


def test_wave_per_position_kernel(ms, oracle_mod, monkeypatch):
    """match_search_wave_kernel (the scan-loop driver's short windows: one wave per position, 65-ary lower bound),
    reached here through DQ_SEARCH_WAVE=1: windows of consecutive positions, every answer against the oracle."""
    monkeypatch.setenv("DQ_SEARCH_WAVE", "1")
    rng = np.random.default_rng(23)
    cases = []
    for size in (1, 2, 3, 64, 65, 66, 999, 4096, 4226, 20000):           # interval sizes around the 64 / 65 split points
        old = oracle_mod.net_random_bytes(size)
        cases.append((old, edited(rng, old, 5)))
        cases.append((rng.integers(0, 2, size, dtype=np.uint8), rng.integers(0, 2, size + 7, dtype=np.uint8)))
    text = oracle_mod.gen_enwik_like(1_500_000, 9, 64 * 1024)
    cases.append((text, edited(rng, text, 30)))
    cases.append((np.zeros(100_000, np.uint8), np.zeros(70_000, np.uint8)))
    for old, new in cases:
        sa = oracle_mod.divsufsort(old)
        starts = [0] + [int(x) for x in rng.integers(0, new.size + 1, 6)] + [max(0, new.size - 50)]
        for s0 in starts:
            cnt = min(new.size + 1 - s0, int(rng.integers(1, 700)))
            want = oracle_mod.bsdiff_search(old, sa, new, scan0=s0, count=cnt)
            got = ms.Search(sa, old, new, scan0=s0, count=cnt)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), (old.size, new.size, s0, cnt)
    old, new = cases[-2]
    sa64 = oracle_mod.divsufsort(old).astype(np.int64)
    want = oracle_mod.bsdiff_search(old, sa64, new, scan0=1000, count=500)
    got = ms.Search(sa64, old, new, scan0=1000, count=500)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])


def test_prefix_table_ranges_and_the_last_suffixes(backend_lib, oracle_mod, monkeypatch):
    """The scan-loop driver starts its searches from a prefix table (2 or 3 bytes): [ptab[v], ptab[v + 1]) also holds,
    at its upper end, the suffixes SHORTER than the prefix that sort between the two patterns (old ending in byte c: the
    one-byte suffix "c" lies in the range of (c - 1, 255)).  Every position, both kernels, with the table forced on
    (DQ_SEARCH_PTAB); queries built to start with exactly those prefixes.  Found by tests/manual/stress_bsdiff.py
    (seed 373, pair 7459: a pre-existing slip of round 2), kept in tests/golden/regress/."""
    import os
    from conftest import GOLDEN_DIR
    from deltaq_amd import Diff, HipMatchSearch
    ms = HipMatchSearch(0)
    rng = np.random.default_rng(3)
    pairs = []
    for tail in ([239], [7, 0], [200, 255], [0], [255], [9, 9]):
        c = tail[0]
        body = rng.integers(0, 256, 70_000, dtype=np.uint8)
        for k in range(0, 60_000, 997):                          # plant the critical prefixes in old, followed by noise
            body[k:k + 3] = [(c - 1) % 256, 255, 255]
            body[k + 500:k + 503] = [c, tail[1] if len(tail) > 1 else 0, 0]
        old = np.concatenate([body, np.array(tail, np.uint8)])
        new = rng.integers(0, 256, 20_000, dtype=np.uint8)
        for k in range(0, 19_000, 211):                          # ... and queries that start with them
            new[k:k + 3] = [(c - 1) % 256, 255, int(rng.integers(0, 256))]
            new[k + 100:k + 102] = [(c - 1) % 256, 255]
            new[k + 150:k + 153] = [c, tail[1] if len(tail) > 1 else 0, 0]
        new[-2:] = [(c - 1) % 256, 255]
        pairs.append((old, new))
    pairs.append((np.load(os.path.join(GOLDEN_DIR, "regress", "bsdiff_373_7459_old.npy")),
                  np.load(os.path.join(GOLDEN_DIR, "regress", "bsdiff_373_7459_new.npy"))))
    for old, new in pairs:
        sa = oracle_mod.divsufsort(old)
        wp, wl = oracle_mod.bsdiff_search(old, sa, new, scan0=0, count=new.size)
        for tab in ("2", "3"):
            monkeypatch.setenv("DQ_SEARCH_PTAB", tab)
            p, l = ms.Search(sa, old, new, scan0=0, count=new.size)
            assert np.array_equal(p, wp) and np.array_equal(l, wl), ("lane kernel", tab, old.size)
            monkeypatch.setenv("DQ_SEARCH_WAVE", "1")
            for s0 in range(0, new.size, 4096):
                cnt = min(4096, new.size - s0)
                p, l = ms.Search(sa, old, new, scan0=s0, count=cnt)
                assert np.array_equal(p, wp[s0:s0 + cnt]) and np.array_equal(l, wl[s0:s0 + cnt]), ("wave kernel", tab, old.size, s0)
            monkeypatch.delenv("DQ_SEARCH_WAVE")
            monkeypatch.delenv("DQ_SEARCH_PTAB")
        ctrl, diff, extra, _ = Diff.Scan(old, new)
        wc, wd, we, _ = oracle_mod.bsdiff_scan(old, sa, new)
        assert np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we), old.size
