"""GPU tests of Diff.Create natively (dq_bsdiff_create / dq_bsdiff_scan_i32): the raw streams of the scan loop --
driven by windows of device match-search answers -- against the oracle's restatement of Diff.cs:91-232 bit for
bit; the container read by a reference-style reader (header + libbz2 + the oracle's ApplyInternal) and by the
product's own Patch.Apply; patches framed by libbz2 applied by the product."""
import bz2
import io
import os

import numpy as np
import pytest

from test_bz2_container import packed, reference_style_patch
from test_gpu_match_search import edited

pytestmark = pytest.mark.gpu


def unpacked(b):
    y = int.from_bytes(b[:7] + bytes([b[7] & 0x7F]), "little")
    return -y if b[7] & 0x80 else y


def reference_style_apply(oracle_mod, old, patch):
    """Patch.Apply as the reference does it: header, three bzip2 streams (libbz2), ApplyInternal (oracle)."""
    assert patch[:8] == b"BSDIFF40"
    lc, ld, newsize = unpacked(patch[8:16]), unpacked(patch[16:24]), unpacked(patch[24:32])
    c = bz2.decompress(patch[32:32 + lc])
    d = bz2.decompress(patch[32 + lc:32 + lc + ld])
    e = bz2.decompress(patch[32 + lc + ld:])
    ctrl = np.array([unpacked(c[i:i + 8]) for i in range(0, len(c), 8)], dtype=np.int64).reshape(-1, 3)
    return oracle_mod.bspatch_apply(old, ctrl, np.frombuffer(d, np.uint8), np.frombuffer(e, np.uint8), newsize)


def pairs(oracle_mod):
    rng = np.random.default_rng(17)
    out = []
    for size in (0, 1, 2, 512, 999, 1024, 4096):                         # BsDiffTests.cs sizes
        old = oracle_mod.net_random_bytes(size)
        out.append((old, old.copy()))
        out.append((old, oracle_mod.gen_uniform(max(size, 3), 9)))
    old = oracle_mod.gen_enwik_like(300_000, 3, 8192)
    out.append((old, edited(rng, old, 30)))                               # text with edits: many short and long matches
    old = oracle_mod.gen_uniform(500_000, 21)
    out.append((old, edited(rng, old, 12)))                               # random data with edits: long matches, jumps
    out.append((old, oracle_mod.gen_uniform(200_000, 22)))                # nothing in common: one Search per byte
    out.append((np.zeros(100_000, np.uint8), np.zeros(90_000, np.uint8)))
    z = np.zeros(2_600_000, np.uint8)                                     # megabytes of equal text: every probe of the exact
    out.append((z, z.copy()))                                             # search is a long comparison (binary probing, slow path)
    z = np.zeros(60_000, np.uint8)                                        # a run with an edit inside: the loop walks on, every
    out.append((z, np.concatenate([z[:20_000], np.arange(1, 7, dtype=np.uint8), z[20_000:]])))   # position far-matching (exact windows)
    per = np.tile(oracle_mod.gen_uniform(1000, 8), 1500)
    out.append((per, np.concatenate([per[:700_000], per[3:]])))           # periodic text with one deletion
    old = oracle_mod.gen_enwik_like(5_000_000, 4, 16384)                  # >= 4 MiB: the search starts from a 3-byte prefix table
    out.append((old, edited(rng, old, 300)))
    new = oracle_mod.gen_uniform(60_000, 23)
    new[1000:1003] = old[77:80]
    out.append((old, new))
    out.append((oracle_mod.gen_uniform(1000, 1), np.zeros(0, np.uint8)))
    return out


# the anchor search of the scan loop runs on the device by default (dq_anchor_scan.h: one persistent launch per new
# file); the host loop over windows of device answers stays as the path a starved launch falls back to
# (several grids on one new file -- "chains", dq_diff.hip -- start from 256 KiB of new by default: forced onto the small
# pairs here, 4 grids of 64 workgroups from 4 KiB on, and 8 grids of 16 that leave after one iteration end behind the next
# grid's start and after one window of one position per lane)
SCAN_PATHS = [{}, {"DQ_SCAN_DEVICE": "0"}, {"DQ_SCAN_GROUPS": "8"}, {"DQ_SCAN_GROUPS": "48"},
              {"DQ_SCAN_CHAINS": "4", "DQ_SCAN_MIN_SEG": "2048"},
              {"DQ_SCAN_CHAINS": "5", "DQ_SCAN_MIN_SEG": "1000", "DQ_SCAN_PAR_EMIT": "0"},       # (every triple computed on the calling thread)
              {"DQ_SCAN_CHAINS": "8", "DQ_SCAN_MIN_SEG": "300", "DQ_SCAN_GROUPS": "16", "DQ_SCAN_EXTRA": "1", "DQ_SCAN_LANE_BUDGET": "1"}]


@pytest.mark.parametrize("env", SCAN_PATHS, ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "device-scan")
def test_raw_streams_equal_the_reference_loop(backend_lib, oracle_mod, monkeypatch, env):
    from deltaq_amd import Diff
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for old, new in pairs(oracle_mod):
        ctrl, diff, extra, stats = Diff.Scan(old, new)
        sa = oracle_mod.divsufsort(old)
        wc, wd, we, searches = oracle_mod.bsdiff_scan(old, sa, new)
        assert np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we), (old.size, new.size)
        assert stats["searches"] == searches


def test_chains_are_joined_between_similar_files_and_dropped_between_unrelated_ones(backend_lib, oracle_mod, monkeypatch):
    """Several grids on one new file (dq_diff.hip, "chains"): between similar files every speculative grid is joined --
    same place, same shift -- and most triples come from the grids' own emitter threads; between unrelated files the
    grids leave after their budget of lane windows and are dropped; the raw streams and the Search count are the
    reference loop's either way.  By default the grids start at 256 KiB of new (128 KiB each)."""
    from deltaq_amd import Diff, _abi
    rng = np.random.default_rng(23)
    old = oracle_mod.gen_uniform(6_000_000, 31)
    similar = edited(rng, old, 400)
    text = oracle_mod.gen_enwik_like(5_000_000, 9, 16384)
    for o, x, kind in ((old, similar, "similar"), (text, edited(rng, text, 300), "similar"),
                       (old, oracle_mod.gen_uniform(3_000_000, 32), "unrelated"), (old, old[1_000_000:5_500_000].copy(), "similar")):
        wc, wd, we, searches = oracle_mod.bsdiff_scan(o, oracle_mod.divsufsort(o), x)
        for env in ({}, {"DQ_SCAN_PAR_EMIT": "0"}, {"DQ_SCAN_CHAINS": "3"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ctrl, diff, extra, stats = Diff.Scan(o, x)
            info = _abi.last_diff_info()
            for k in env:
                monkeypatch.delenv(k)
            assert np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we) and stats["searches"] == searches, (kind, env)
            assert info["host_loop_fallbacks"] == 0 and info["chains_launched"] >= 2, (kind, env, info)
            if kind == "similar":
                assert info["chains_joined"] >= 1 and info["chains_joined"] + info["chains_dropped"] <= info["chains_launched"], (env, info)
                if "DQ_SCAN_PAR_EMIT" in env:
                    assert info["triples_from_chain_emitters"] == 0, info
                elif x.size > 4_000_000 and ctrl.size >= 300:
                    assert info["triples_from_chain_emitters"] > ctrl.size // 3 // 4, (env, info, ctrl.shape)
            else:
                assert info["chains_joined"] == 0, (env, info)
    # one grid alone below 256 KiB of new
    small = edited(rng, old[:200_000], 20)
    Diff.Scan(old, small)
    info = _abi.last_diff_info()
    assert info["chains_launched"] == 1 and info["chains_joined"] == 0 and info["scan_groups"] == 128, info


def test_silent_iteration_ends_and_a_full_list(backend_lib, oracle_mod, monkeypatch):
    """One byte changed every 60 bytes of a 4 MiB file: the loop ends ~70 000 iterations without a triple (the old
    alignment explains every match) -- grids are joined at such ends by place AND shift -- and one grid alone writes more
    entries than its list holds per launch (32 768): it leaves, is read to the end and launched again from where it stood."""
    from deltaq_amd import Diff, _abi
    old = oracle_mod.gen_uniform(4 << 20, 91)
    new = old.copy()
    new[100::60] ^= 0x5a
    wc, wd, we, searches = oracle_mod.bsdiff_scan(old, oracle_mod.divsufsort(old), new)
    for env in ({"DQ_SCAN_CHAINS": "1"}, {"DQ_SCAN_CHAINS": "2"}, {}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctrl, diff, extra, stats = Diff.Scan(old, new)
        info = _abi.last_diff_info()
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we) and stats["searches"] == searches, env
        assert info["host_loop_fallbacks"] == 0, info
        if env.get("DQ_SCAN_CHAINS") == "1":
            assert info["chains_launched"] >= 3 and info["chains_joined"] == 0, info          # (its list was full twice)
        elif env:
            assert info["chains_joined"] >= 1 and info["chains_launched"] >= 3, info
        else:
            assert info["chains_joined"] >= 8, info


def test_create_apply_roundtrip_and_cross_compatibility(backend_lib, oracle_mod):
    from deltaq_amd import Diff, Patch, HipSuffixSort
    for old, new in pairs(oracle_mod):
        out = io.BytesIO()
        Diff.Create(old, new, out, HipSuffixSort(0))
        patch = out.getvalue()
        assert patch[:8] == b"BSDIFF40" and len(patch) <= backend_lib.dq_bsdiff_patch_bound(old.size, new.size)
        assert Patch.Apply(old, patch) == new.tobytes()                                   # product reads product
        assert np.array_equal(reference_style_apply(oracle_mod, old, patch), new)         # a reference-style reader reads product
        assert Patch.Apply(old, reference_style_patch(oracle_mod, old, new)) == new.tobytes()     # product reads reference-style
    with pytest.raises(ValueError):
        Diff.Create(b"a", b"b", None)


def test_bzip2_blocks_through_the_device_sorter(backend_lib, oracle_mod):
    """Streams longer than one 900 kB bzip2 block: every block's Burrows-Wheeler transform is a run of the HIP sorter
    on block+block (1.8 MB).  Unrelated files of 3 MB make the extra stream 3 MB long."""
    from deltaq_amd import Diff, Patch
    old = oracle_mod.gen_uniform(1_000_000, 5)
    new = oracle_mod.gen_enwik_like(3_000_000, 6, 4096)
    patch = Diff.CreateBytes(old, new)
    assert np.array_equal(reference_style_apply(oracle_mod, old, patch), new)
    assert Patch.Apply(old, patch) == new.tobytes()
    assert len(patch) < new.size // 2                                     # text compresses: the framing does its job


def test_framing_behind_the_scan_writes_the_same_patch(backend_lib, oracle_mod, monkeypatch):
    """The diff and extra streams are framed while the scan loop still appends to them (bz2::StreamEncoder behind the
    emitter, full blocks encoded on their own threads): byte for byte the patch of framing the finished streams."""
    from deltaq_amd import Diff, Patch
    rng = np.random.default_rng(17)
    big_old = oracle_mod.gen_enwik_like(4_000_000, 12, 16384)
    cases = list(pairs(oracle_mod)) + [(oracle_mod.gen_uniform(1_000_000, 5), oracle_mod.gen_enwik_like(3_000_000, 6, 4096)),
                                       (big_old, edited(rng, big_old, 300)), (big_old, big_old.copy())]
    monkeypatch.setenv("DQ_FRAME_AFTER", "1")
    want = [Diff.CreateBytes(o, x) for o, x in cases]
    monkeypatch.delenv("DQ_FRAME_AFTER")
    for follow_min in ("0", None):
        if follow_min is None:
            monkeypatch.delenv("DQ_FRAME_FOLLOW_MIN")
        else:
            monkeypatch.setenv("DQ_FRAME_FOLLOW_MIN", follow_min)
        for (o, x), w in zip(cases, want):
            p = Diff.CreateBytes(o, x)
            assert p == w, (o.size, x.size, follow_min)
    for (o, x), w in zip(cases, want):
        assert Patch.Apply(o, w) == x.tobytes()


def test_regression_pairs_found_by_the_stress_runs(backend_lib, oracle_mod, monkeypatch):
    """Pairs tests/manual/stress_bsdiff.py caught a build on (tests/golden/regress/bsdiff_*_{old,new}.npy).
    bsdiff_1120044142_236: windows of one position per lane and of one per wave shared their two answer buffers -- a slot
    beyond the 512 of a wave window kept the word of the last lane window, and its two-bit tag came round again after
    three uses of the buffer (raw streams differed in two runs out of three)."""
    import glob
    from conftest import GOLDEN_DIR
    from deltaq_amd import Diff
    olds = sorted(glob.glob(os.path.join(GOLDEN_DIR, "regress", "bsdiff_*_old.npy")))
    assert olds
    for f in olds:
        old, new = np.load(f), np.load(f.replace("_old.npy", "_new.npy"))
        wc, wd, we, _ = oracle_mod.bsdiff_scan(old, oracle_mod.divsufsort(old), new)
        for env in SCAN_PATHS:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for rep in range(6):
                ctrl, diff, extra, _ = Diff.Scan(old, new)
                assert np.array_equal(ctrl, wc) and np.array_equal(diff, wd) and np.array_equal(extra, we), (os.path.basename(f), env, rep)
            for k in env:
                monkeypatch.delenv(k)


def test_one_old_many_new_index(backend_lib, oracle_mod):
    """DiffIndex: the old file is sorted once; every Create returns the patch Diff.Create writes for that pair.
    Also built on device-resident (text, suffix array) tensors of the caller, as a broadcast receiver holds them."""
    import torch
    from deltaq_amd import Diff, DiffIndex, HipSuffixSort, Patch
    rng = np.random.default_rng(5)
    for old in (oracle_mod.gen_enwik_like(5_000_000, 4, 16384), oracle_mod.gen_uniform(300_000, 21),
                oracle_mod.net_random_bytes(4096), np.zeros(0, np.uint8)):
        news = [edited(rng, old, k) for k in (3, 40)] if old.size else []
        news += [old.copy(), oracle_mod.gen_uniform(50_000, 3), np.zeros(0, np.uint8)]
        want = [Diff.CreateBytes(old, x) for x in news]
        with DiffIndex(old, 0) as ix:
            for x, w in zip(news, want):
                p = ix.Create(x)
                assert p == w
                assert Patch.Apply(old, p) == x.tobytes()
        if old.size > 8192:
            dT = torch.from_numpy(old).cuda()
            dSA = HipSuffixSort(0).Sort(dT)
            with DiffIndex(old, 0, device_text=dT, device_sa=dSA) as ix:
                assert [ix.Create(x) for x in news] == want


def test_index_clone_gives_the_same_patches(backend_lib, oracle_mod):
    """dq_bsdiff_index_clone: one more copy of an index -- on the same device and, where the node has more, on every other
    one -- by device-to-device copies of text, suffix array and prefix table.  Every clone must return the patches
    dq_bsdiff_create returns, go on working after the source is gone, and be freed on its own."""
    import threading
    import torch
    from deltaq_amd import Diff, DiffIndex, HipSuffixSort, Patch
    rng = np.random.default_rng(8)
    ndev = backend_lib.dq_device_count()
    for old in (oracle_mod.gen_enwik_like(5_000_000, 6, 16384),        # 3-byte prefix table
                oracle_mod.gen_uniform(300_000, 22),                   # 2-byte
                oracle_mod.net_random_bytes(4096)):                    # none
        news = [edited(rng, old, k) for k in (2, 30)] + [oracle_mod.gen_uniform(40_000, 5), np.zeros(0, np.uint8)]
        want = [Diff.CreateBytes(old, x) for x in news]
        src = DiffIndex(old, 0)
        clones = [src.clone(d) for d in range(ndev)] + [src.clone(0)]
        assert [src.Create(x) for x in news] == want
        src.close()                                                    # the copies own what they hold
        got = [None] * len(clones)

        def work(k):
            got[k] = [clones[k].Create(x) for x in news]
        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(clones))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for k, g in enumerate(got):
            assert g == want, f"clone {k} of {len(clones)} (devices: {ndev})"
        assert Patch.Apply(old, got[0][0]) == news[0].tobytes()
        for c in clones:
            c.close()
    # a clone of an index that sits on the caller's buffers (a broadcast receiver's) owns copies of them
    old = oracle_mod.gen_uniform(200_000, 23)
    dT = torch.from_numpy(old).cuda()
    dSA = HipSuffixSort(0).Sort(dT)
    with DiffIndex(old, 0, device_text=dT, device_sa=dSA) as ix:
        c = ix.clone(0)
    del dT, dSA
    torch.cuda.empty_cache()
    x = edited(rng, old, 10)
    assert c.Create(x) == Diff.CreateBytes(old, x)
    c.close()
    with pytest.raises(Exception):
        DiffIndex(old, 0).clone(ndev + 3)                              # no such device


def test_concurrent_callers_on_one_device(backend_lib, oracle_mod):
    """Four threads on one device at once: dq_bsdiff_create, diffs against a shared index, batched match searches
    and plain sorts -- the scan-loop windows (polled pinned answers, second-stage mailbox, cached buffers) are one
    per device, so callers must take turns without mixing their answers up."""
    import threading
    from deltaq_amd import Diff, DiffIndex, HipMatchSearch, HipSuffixSort, Patch
    rng = np.random.default_rng(11)
    old = oracle_mod.gen_enwik_like(2_000_000, 9, 16384)
    news = [edited(rng, old, 25 + 5 * k) for k in range(4)]
    want = [Diff.CreateBytes(old, x) for x in news]
    sa = oracle_mod.divsufsort(old)
    scans = np.sort(rng.integers(0, news[0].size, 20000)).astype(np.int64)
    pos_w, len_w = oracle_mod.bsdiff_search(old, sa, news[0], scans)
    index = DiffIndex(old, 0)
    errs = []

    def create(k):
        for _ in range(3):
            if Diff.CreateBytes(old, news[k]) != want[k]:
                errs.append(("create", k))

    def via_index(k):
        for _ in range(3):
            if index.Create(news[k]) != want[k]:
                errs.append(("index", k))

    def searches():
        ms = HipMatchSearch(0)
        for _ in range(3):
            p, l = ms.Search(sa, old, news[0], scans)
            if not (np.array_equal(p, pos_w) and np.array_equal(l, len_w)):
                errs.append(("search",))

    def sorts():
        s = HipSuffixSort(0)
        for _ in range(3):
            if not np.array_equal(s.Sort(old), sa):
                errs.append(("sort",))

    def guarded(f, *a):
        try:
            f(*a)
        except Exception as e:                                            # noqa: BLE001
            errs.append((f.__name__, repr(e)))

    ths = [threading.Thread(target=guarded, args=(create, 0)), threading.Thread(target=guarded, args=(via_index, 1)),
           threading.Thread(target=guarded, args=(via_index, 2)), threading.Thread(target=guarded, args=(create, 3)),
           threading.Thread(target=guarded, args=(searches,)), threading.Thread(target=guarded, args=(sorts,))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    index.close()
    assert not errs, errs
    assert Patch.Apply(old, want[2]) == news[2].tobytes()
