"""GPU parity tests: the HIP path, called through the C ABI, against the oracle.

They mirror the reference's own provider tests (LibDivSufSortTests.cs): shruggy string,
every asset file, the seeded random buffers -- checked by the restated Verify
(strict order + sufcheck) AND bit-compared with the oracle / golden digests -- plus the
edge cases a doubling algorithm is sensitive to."""
import hashlib

import numpy as np
import pytest

from conftest import asset_names, load_asset

pytestmark = pytest.mark.gpu


def sha_i32(sa):
    return hashlib.sha256(np.asarray(sa).astype("<i4").tobytes()).hexdigest()


@pytest.fixture(scope="module")
def ldss(backend_lib):
    from deltaq_amd import HipSuffixSort
    assert backend_lib.dq_device_count() >= 1, "no MI355X visible: the HIP path cannot be tested"
    return HipSuffixSort(0)


@pytest.fixture(params=["auto", "device-wide"])
def pipeline(request, monkeypatch):
    """Texts of n <= 8192 bytes are sorted by the single-workgroup kernel (dq_small.h); DQ_SMALL_N=0
    sends them down the device-wide pipeline instead.  The fixture-scale tests run both ways."""
    if request.param == "device-wide":
        monkeypatch.setenv("DQ_SMALL_N", "0")
    return request.param


def Verify(oracle_mod, T, SA):
    """LibDivSufSortTests.Verify (cs:43-64)."""
    oracle_mod.verify(T, np.asarray(SA))


def test_CheckShruggy(ldss, oracle_mod, pipeline):
    T = "¯\\_(ツ)_/¯".encode("utf-8")
    SA = ldss.Sort(T)
    Verify(oracle_mod, T, SA)
    assert SA.tolist() == [4, 8, 10, 2, 3, 9, 6, 7, 12, 1, 11, 0, 5]


@pytest.mark.parametrize("name", asset_names())
def test_CheckFile(ldss, oracle_mod, golden, name, pipeline):
    T = load_asset(name)
    SA = ldss.Sort(T)
    Verify(oracle_mod, T, SA)
    assert sha_i32(SA) == golden["assets"][name]["sa_sha256_le_i32"]
    assert np.array_equal(SA, oracle_mod.divsufsort(T))


@pytest.mark.parametrize("size", [0, 1, 2, 4, 8, 16, 32, 51, 0x1000, 0x8000, 0x8000 - 1])
def test_CheckRandomBuffer(ldss, oracle_mod, golden, size, pipeline):
    T = oracle_mod.net_random_bytes(size)
    SA = np.zeros(size, dtype=np.int32)             # AllocationMode.Clear, cs:143
    ldss.Sort(T, SA)
    Verify(oracle_mod, T, SA)
    assert sha_i32(SA) == golden["net_random_670761"][str(size)]["sa_sha256_le_i32"]


def test_caller_buffer_may_hold_garbage_and_sentinel_slot_is_untouched(ldss, oracle_mod):
    # Diff.Create passes I[..^1] of an (n+1)-int buffer (Diff.cs:78,89-90): I[n] must stay 0
    T = oracle_mod.net_random_bytes(5000)
    I = np.full(T.size + 1, 0x5A5A5A5A, dtype=np.int32)
    I[-1] = 0
    ldss.Sort(T, I[:-1])
    assert I[-1] == 0
    assert np.array_equal(I[:-1], oracle_mod.divsufsort(T))


def pathological_cases(oracle_mod):
    out = {}
    for n in (3, 7, 8, 9, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 10000):
        out[f"zeros{n}"] = np.zeros(n, np.uint8)
        out[f"ff{n}"] = np.full(n, 255, np.uint8)
        out[f"ab{n}"] = np.tile(np.array([97, 98], np.uint8), n)[:n]
        out[f"akb{n}"] = np.concatenate([np.full(n, 97, np.uint8), [98]]).astype(np.uint8)
        out[f"rnd{n}+zeros"] = np.concatenate([oracle_mod.net_random_bytes(n), np.zeros(9, np.uint8)])
        out[f"rnd{n}"] = oracle_mod.net_random_bytes(n)
    a, b = b"a", b"ab"
    while len(b) < 20000:
        a, b = b, b + a
    out["fibonacci"] = np.frombuffer(b, dtype=np.uint8)
    out["thue-morse"] = np.array([bin(i).count("1") & 1 for i in range(1 << 14)], dtype=np.uint8)
    out["zero-tail"] = np.array([5, 0, 0, 5, 0, 7, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 5, 0], np.uint8)
    out["4symbols"] = oracle_mod.gen_uniform(100000, 7) & 3
    out["period1000"] = np.tile(oracle_mod.gen_uniform(1000, 9), 60)
    return out


def test_pathological_inputs(ldss, oracle_mod, pipeline):
    for name, T in pathological_cases(oracle_mod).items():
        T = np.ascontiguousarray(T, dtype=np.uint8)
        SA = ldss.Sort(T)
        ref = oracle_mod.divsufsort(T)
        assert np.array_equal(SA, ref), name


def test_single_workgroup_sorter_every_small_size(ldss, oracle_mod, backend_lib):
    """dq_small.h: every length 1..300 plus the sizes around each elements-per-thread step and the
    8192 limit, on alphabets of 1, 2, 4 and 256 symbols, host and device entry points, both index widths."""
    import ctypes
    import torch
    rng = np.random.default_rng(20261001)
    sizes = list(range(1, 301)) + [1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097, 7000, 8191, 8192, 8193]
    for n in sizes:
        sigma = (1, 2, 4, 256)[n % 4]
        T = rng.integers(0, sigma, size=n, dtype=np.uint8) if sigma > 1 else np.full(n, 7, np.uint8)
        if n % 5 == 0:
            T[-min(n, 9):] = 0                    # genuine zero tail against the zero padding
        ref = oracle_mod.divsufsort(T)
        assert np.array_equal(ldss.Sort(T), ref), n
        if n % 7 == 0 or n > 300:
            assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), ref.astype(np.int64)), n
            dSA = ldss.Sort(torch.from_numpy(T).cuda())
            assert np.array_equal(dSA.cpu().numpy(), ref), n
    # the kernel really ran for these sizes: its profile category counts the launches
    backend_lib.dq_profile_enable(1)
    backend_lib.dq_profile_reset()
    ldss.Sort(oracle_mod.net_random_bytes(5000))
    launches = ctypes.c_int64()
    backend_lib.dq_profile_get(14, ctypes.byref(launches), None, None, None)     # DQ_K_SMALL_SORT
    backend_lib.dq_profile_enable(0)
    assert launches.value == 1


def test_tie_bits_of_the_last_pass(ldss, oracle_mod, backend_lib):
    """Packed sorts record ties in the last digit pass (dq_ties.h): pairs inside a tile, pairs that
    straddle tiles (any n > one tile has them), runs continuing over several 64-bit words, and the
    fallback when a run of equal keys is too long for the per-thread walk."""
    rnd = oracle_mod.gen_uniform
    cases = {
        "random 3M": rnd(3_000_000, 21),
        "run of 1000 zeros": np.concatenate([rnd(700_000, 22), np.zeros(1000, np.uint8), rnd(400_000, 23)]),
        "run of 60000 x 0x41 (fallback)": np.concatenate([rnd(2_000_000, 24), np.full(60_000, 0x41, np.uint8),
                                                          rnd(1_000_000, 25)]),
        "repeat of 5000 bytes": np.concatenate([rnd(900_000, 26), rnd(900_000, 26)[1000:6000], rnd(300_000, 27)]),
        "exactly one tile": rnd(12288, 28), "one tile + 1": rnd(12289, 29), "64 KiB": rnd(1 << 16, 30),
        "zero tail": np.concatenate([rnd(500_000, 31), np.zeros(9, np.uint8)]),
    }
    for name, T in cases.items():
        T = np.ascontiguousarray(T, dtype=np.uint8)
        assert np.array_equal(ldss.Sort(T), oracle_mod.divsufsort(T)), name
    T = cases["random 3M"]
    assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), oracle_mod.divsufsort(T).astype(np.int64))


def test_pair_chains(ldss, oracle_mod, backend_lib, monkeypatch):
    """dq_pair_chains.h: tied pairs inside long repeats take the answer of the end of their chain.  Taken on its own
    when a doubling round leaves most of a LONG list tied (>= 2^23 entries: shorter ones are cheaper to keep
    doubling since the LDS class); here with the default triggers on lists of any length (DQ_PAIR_CHAINS_MIN=1), then
    forced before and after every round."""
    import ctypes
    rnd = oracle_mod.gen_uniform

    def launches(T, dtype=np.int32):
        backend_lib.dq_profile_reset()
        backend_lib.dq_profile_enable(1)
        sa = ldss.Sort(T, index_dtype=dtype)
        backend_lib.dq_profile_enable(0)
        n = ctypes.c_int64()
        backend_lib.dq_profile_get(17, ctypes.byref(n), None, None, None)        # DQ_K_PAIR_CHAINS
        assert np.array_equal(sa, oracle_mod.divsufsort(T).astype(dtype))
        return n.value

    assert launches(oracle_mod.gen_enwik_like(6_000_000, 53, 65536)) == 0                     # a short list: no phase by default
    monkeypatch.setenv("DQ_PAIR_CHAINS_MIN", "2048")
    base = rnd(3_000_000, 51)
    one_copy = np.concatenate([base, base[100_000:160_000], rnd(500_000, 52)])               # pairs only: one chain of 60000
    assert launches(one_copy) >= 2                                                            # split + link phases ran
    text = oracle_mod.gen_enwik_like(6_000_000, 53, 65536)                                    # ~90 copies of up to 64 KiB
    assert launches(text) >= 2
    assert launches(rnd(2_000_000, 54)) == 0                                                  # nothing stagnates
    doubled = oracle_mod.gen_enwik_like(2_500_000, 55, 1 << 20)
    doubled = np.concatenate([doubled, rnd(100_000, 56), doubled, rnd(50_000, 57)])          # a whole file twice: more than n/2
    assert launches(doubled) >= 2                                                             # suffixes tied, the records still fit
    monkeypatch.delenv("DQ_PAIR_CHAINS_MIN")
    for force in ("1", "2"):
        monkeypatch.setenv("DQ_PAIR_CHAINS", force)
        monkeypatch.setenv("DQ_SMALL_N", "0")
        x = rnd(200_000, 55)
        cases = {
            "copy at the very end (b+1 = n)": np.concatenate([rnd(300_000, 56), x, rnd(100_000, 57), x]),
            "three copies (blocked chains)": np.concatenate([x, rnd(1000, 58), x, rnd(1000, 59), x[:150_000], rnd(5, 60)]),
            "overlapping copies": np.concatenate([x, x[50_000:], x[100_000:], x[:30_000]]),
            "periodic": np.tile(rnd(4099, 61), 100),
            "square of a random word": np.concatenate([x, x]),
            "two symbols": rnd(700_000, 62) & 1,
            "all equal": np.full(150_000, 9, np.uint8),
            "text": oracle_mod.gen_enwik_like(900_000, 63, 8192),
            "short": rnd(3000, 64) & 3,
        }
        for name, T in cases.items():
            T = np.ascontiguousarray(T, dtype=np.uint8)
            assert np.array_equal(ldss.Sort(T), oracle_mod.divsufsort(T)), (force, name)
        T = np.ascontiguousarray(cases["three copies (blocked chains)"])
        assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), oracle_mod.divsufsort(T).astype(np.int64))


def test_coded_round0(ldss, oracle_mod, backend_lib, monkeypatch):
    """dq_alpha_code.h / dq_coded_keys.h: text-like inputs of >= 8 MiB sort 64-bit keys made of alphabetic codewords
    (12...14 characters instead of 8 bytes).  The path is recognisable by the ties it leaves; it is forced here on
    small inputs and on alphabets where it does not pay (fixed 8-bit codewords = raw bytes)."""

    from deltaq_amd import _abi

    def tied_after_round0(T, dtype=np.int32):
        sa = ldss.Sort(T, index_dtype=dtype)
        info = _abi.last_sort_info()
        assert np.array_equal(sa, oracle_mod.divsufsort(T).astype(dtype))
        return info["initial_active"]

    text = oracle_mod.gen_enwik_like(9_000_001, 21, 65536)
    coded = tied_after_round0(text)
    monkeypatch.setenv("DQ_CODED", "0")
    plain = tied_after_round0(text)
    monkeypatch.delenv("DQ_CODED")
    assert coded * 4 < plain * 3, (coded, plain)                         # 12-14 characters per key instead of 8: far fewer ties
    # real text: this repository's own documents, repeated with scattered edits up to the size where the path engages
    import glob, os
    from conftest import ROOT
    docs = b"".join(open(f, "rb").read() for f in sorted(glob.glob(os.path.join(ROOT, "*.md"))))
    rng = np.random.default_rng(5)
    real = np.frombuffer(docs * (9_000_000 // max(len(docs), 1) + 1), dtype=np.uint8)[:9_500_000].copy()
    edits = rng.integers(0, real.size, 20_000)
    real[edits] = rng.integers(32, 127, edits.size).astype(np.uint8)
    tied_after_round0(real)
    dna = (oracle_mod.gen_uniform(10_000_000, 23) & 3) + 65
    tied_after_round0(np.ascontiguousarray(dna, dtype=np.uint8))         # 4 symbols: fixed 4-bit codewords, 16 characters per key
    monkeypatch.setenv("DQ_CODED", "1")
    monkeypatch.setenv("DQ_PACKED", "0")
    monkeypatch.setenv("DQ_KEY_BYTES", "8")
    monkeypatch.setenv("DQ_SMALL_N", "0")
    rnd = oracle_mod.gen_uniform
    cases = {
        "uniform bytes (8-bit codewords)": rnd(300_000, 31),
        "text": oracle_mod.gen_enwik_like(700_003, 32, 4096),
        "one dominant byte": np.where(rnd(400_000, 33) < 250, 0x20, rnd(400_000, 34)),
        "zeros present + zero tail": np.concatenate([rnd(200_000, 35) & 7, np.zeros(50, np.uint8)]),
        "all zeros": np.zeros(100_000, np.uint8),
        "all 0xff": np.full(70_001, 0xFF, np.uint8),
        "two symbols": (rnd(500_000, 36) & 1) * 200,
        "periodic": np.tile(rnd(997, 37), 300),
        "129 symbols": rnd(600_000, 38) % 129,
        "tiny": rnd(67, 39) & 15,
    }
    for name, T in cases.items():
        T = np.ascontiguousarray(T, dtype=np.uint8)
        assert np.array_equal(ldss.Sort(T), oracle_mod.divsufsort(T)), name
    T = np.ascontiguousarray(oracle_mod.gen_enwik_like(1_000_001, 40, 8192))
    assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), oracle_mod.divsufsort(T).astype(np.int64))


def test_bucketed_round0(ldss, oracle_mod, backend_lib, monkeypatch):
    """dq_bucket_sort.h: two digit passes on the top 16 key bits, then every bucket finished in LDS.  Taken by
    random-like inputs of >= 12 MiB on its own; forced here on small and on unsuitable inputs, where tiles
    spanning too many buckets, over-long buckets and over-full bins must fall back to the plain passes."""
    import ctypes
    rnd = oracle_mod.gen_uniform

    def bucket_launches(T):
        backend_lib.dq_profile_reset()
        backend_lib.dq_profile_enable(1)
        sa = ldss.Sort(T)
        backend_lib.dq_profile_enable(0)
        n = ctypes.c_int64()
        backend_lib.dq_profile_get(15, ctypes.byref(n), None, None, None)        # DQ_K_BUCKET_SORT
        assert np.array_equal(sa, oracle_mod.divsufsort(T))
        return n.value

    assert bucket_launches(rnd(13_000_000, 1)) == 2                     # bounds + sort kernel
    assert bucket_launches(rnd((17 << 20) + 13, 2)) == 2
    assert bucket_launches(rnd(5_000_000, 3)) == 0                      # too small: tiles would span too many buckets
    T = rnd(16_000_000, 9)
    T[::7] = 0                                                          # skewed: the order-0 model says no
    assert bucket_launches(T) == 0
    T = rnd(14_000_000, 10)
    T[5_000_003:5_000_203] = 0                                          # zero padding, 200 bytes: seen by the histogram pass
    assert bucket_launches(T) == 0
    monkeypatch.setenv("DQ_BUCKET", "1")
    for n in (70_000, 300_001, 1 << 20, 3_000_000):
        assert bucket_launches(rnd(n, n)) == 2
    cases = {
        "6-bit alphabet": rnd(2_000_000, 5) & 0x3F,
        "run of 60000 x 0x41": np.concatenate([rnd(1_000_000, 24), np.full(60_000, 0x41, np.uint8), rnd(5_000_000, 25)]),
        "repeat + zero tail": np.concatenate([rnd(3_000_000, 26), rnd(3_000_000, 26)[1000:6000], np.zeros(9, np.uint8)]),
        "all zeros": np.zeros(200_000, np.uint8),
        "two symbols": rnd(1_000_000, 3) & 1,
    }
    for name, T in cases.items():
        T = np.ascontiguousarray(T, dtype=np.uint8)
        assert np.array_equal(ldss.Sort(T), oracle_mod.divsufsort(T)), name
    T = rnd(4_000_001, 77)
    assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), oracle_mod.divsufsort(T).astype(np.int64))
    # three digit passes + 3-byte buckets (what texts beyond ~300 MiB take, e.g. BASELINE configs[3]), forced on 40 MB
    monkeypatch.setenv("DQ_BUCKET", "3")
    T = rnd(40_000_003, 78)
    T[1_000_000:1_003_000] = T[9_000_000:9_003_000]                      # a repeat: leftovers of the finisher
    backend_lib.dq_profile_reset()
    backend_lib.dq_profile_enable(1)
    sa = ldss.Sort(T)
    backend_lib.dq_profile_enable(0)
    from deltaq_amd import _abi
    assert _abi.profile_snapshot()["bucket_sort_kernel"]["launches"] == 2
    assert _abi.last_sort_info()["initial_active"] < 40_000                # ... and it did not fall back: 36-bit keys leave few ties
    assert np.array_equal(sa, oracle_mod.divsufsort(T))
    # one more byte of key beside every word (what n >= 2^30 takes, where a word has room for 33 key bits only): forced
    # on 16 MB with 26-bit words, with and without the byte -- the byte cuts the ties 256-fold
    monkeypatch.setenv("DQ_BUCKET", "1")
    monkeypatch.setenv("DQ_BUCKET_KEYBITS", "26")
    T = rnd(16_000_001, 79)
    T[2_000_000:2_004_000] = T[11_000_000:11_004_000]
    ref = oracle_mod.divsufsort(T)
    tied = {}
    for ext in ("0", "1"):
        monkeypatch.setenv("DQ_BUCKET_EXT", ext)
        backend_lib.dq_profile_reset()
        backend_lib.dq_profile_enable(1)
        sa = ldss.Sort(T)
        backend_lib.dq_profile_enable(0)
        assert _abi.profile_snapshot()["bucket_sort_kernel"]["launches"] == 2, ext
        tied[ext] = _abi.last_sort_info()["initial_active"]
        assert np.array_equal(sa, ref), ext
        assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), ref.astype(np.int64)), ext
    assert tied["1"] * 50 < tied["0"], tied


from structured_inputs import structured_text          # (shared with tests/manual/stress.py and tools/repro_fuzz.py)


FUZZ_ENVS = [
    {},
    {"DQ_BINNED_ISA": "1", "DQ_SMALL_N": "0"},             # suffix-binned first inverse suffix array on every dense input
    {"DQ_SMALL_N": "0"},                                   # everything through the device-wide pipeline
    {"DQ_NO_FUSED_TIES": "1", "DQ_NO_SMALL": "1", "DQ_TAIL_MAX": "0"},         # general rebucket pass, radix-only doubling rounds
    {"DQ_PACKED": "1", "DQ_KEY_BYTES": "2", "DQ_SMALL_N": "0"},   # tie bits with most suffixes tied
    {"DQ_SPARSE": "1", "DQ_SMALL_N": "0"},                 # finisher + key extension + fallback on dense inputs
    {"DQ_NO_BINNED_ISA": "1", "DQ_NO_CHAIN": "1", "DQ_TAIL_MAX": "0"},         # first ISA by scatter, one host round trip per small-group round
    {"DQ_FORCE_RSHIFT": "1", "DQ_SMALL_N": "0"},           # composite keys carry rank >> 1, true rank read from the ISA (n near 2^32)
    {"DQ_BUCKET": "1", "DQ_SMALL_N": "0"},                 # bucketed round 0 wherever packed words are chosen (+ its fallbacks)
    {"DQ_CODED": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_SMALL_N": "0"},   # round-0 keys from alphabetic codewords
    {"DQ_MID_GROUPS": "0", "DQ_SMALL_N": "0", "DQ_TAIL_MAX": "0"},             # doubling rounds with the small-group kernel + radix passes only
    {"DQ_MID_GROUPS": "256", "DQ_SMALL_N": "0"},           # doubling rounds through the LDS class for groups of up to 256 ...
    {"DQ_MID_GROUPS": "1024", "DQ_SMALL_N": "0", "DQ_PAIR_CHAINS": "0", "DQ_NO_BINNED_ISA": "1", "DQ_TAIL_MAX": "0"},   # ... 1024 members
    {"DQ_RUNS": "1", "DQ_SMALL_N": "0", "DQ_TAIL_MAX": "0"},                   # run lengths + run-order round on every input
    {"DQ_RUNS": "1", "DQ_SMALL_N": "0", "DQ_NO_SMALL": "1", "DQ_NO_BINNED_ISA": "1", "DQ_TAIL_MAX": "0"},
    {"DQ_PAIR_CHAINS": "2", "DQ_SMALL_N": "0"},            # tied pairs decided chain by chain as early and as often as allowed
    {"DQ_PAIR_CHAINS": "2", "DQ_NO_BINNED_ISA": "1", "DQ_SPARSE": "0", "DQ_SMALL_N": "0", "DQ_TAIL_MAX": "0"},
    {"DQ_LATE_RUNS_MIN": "1", "DQ_SMALL_N": "0", "DQ_TAIL_MAX": "0"},          # run lengths + run-order round as soon as large groups stagnate
    {"DQ_LATE_RUNS_MIN": "1", "DQ_SMALL_N": "0", "DQ_MID_GROUPS": "256", "DQ_PAIR_CHAINS": "0", "DQ_TAIL_MAX": "0"},
    {"DQ_LATE_RUNS_MIN": "1", "DQ_SMALL_N": "0", "DQ_RUN_PERIOD": "4", "DQ_PAIR_CHAINS": "0"},   # ... with a fixed period of 4
    {"DQ_UPD_BIN": "2", "DQ_UPD_BIN_MIN": "1", "DQ_SMALL_N": "0", "DQ_TAIL_MAX": "0"},   # rank updates binned by suffix (two passes) before they are applied
    {"DQ_UPD_BIN": "1", "DQ_UPD_BIN_MIN": "1", "DQ_SMALL_N": "0", "DQ_RUNS": "1", "DQ_TAIL_MAX": "0"},   # ... one pass
    {"DQ_NO_UPD_WORDS": "1", "DQ_SMALL_N": "0"},           # rank updates as (rank, suffix) in two arrays
    {"DQ_SPARSE": "1", "DQ_BINNED_ISA": "1", "DQ_SMALL_N": "0"},     # suffix-binned inverse suffix array at the sparse-to-dense switch
    # lists of more than n/2 tied suffixes through the LDS class (third list buffer, round 5): 2-byte keys tie nearly everybody
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_SMALL_N": "0"},
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_SMALL_N": "0", "DQ_BINNED_ISA": "1", "DQ_UPD_BIN_MIN": "1", "DQ_TAIL_MAX": "0"},
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_SMALL_N": "0", "DQ_MID_GROUPS": "0", "DQ_NO_CHAIN": "1"},
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_SMALL_N": "0", "DQ_NO_WIDE_SMALL": "1", "DQ_TAIL_MAX": "0"},   # ... and as before: the radix path
    {"DQ_NO_L_SHIFT": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_SMALL_N": "0", "DQ_MID_GROUPS": "256", "DQ_TAIL_MAX": "0"},   # radix-list keys with the full rank
    {"DQ_UPD_WINDOW": "1", "DQ_SMALL_N": "0", "DQ_TAIL_MAX": "0"},             # rank updates applied span by span inside LDS (isa_update_window_kernel)
    {"DQ_UPD_WINDOW": "1", "DQ_SMALL_N": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_RUNS": "1"},
    {"DQ_UPD_WINDOW": "0", "DQ_SMALL_N": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_TAIL_MAX": "0"},
    # the last rounds of a sort in one launch (dq_tail.h, round 5; default from 4096 tied suffixes down): other switch points, off
    {"DQ_TAIL_MAX": "64", "DQ_SMALL_N": "0"},
    {"DQ_TAIL_MAX": "1000", "DQ_SMALL_N": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
    {"DQ_TAIL_MAX": "4096", "DQ_SMALL_N": "0", "DQ_RUNS": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0"},
    {"DQ_TAIL_MAX": "0", "DQ_SMALL_N": "0"},
    # chained rounds and tail kernel with ONE rank per member and round (before round 5's three: dq_mid_groups.h kSteps)
    {"DQ_CHAIN_STEPS": "1", "DQ_SMALL_N": "0"},
    {"DQ_CHAIN_STEPS": "1", "DQ_TAIL_MAX": "0", "DQ_SMALL_N": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
    {"DQ_CHAIN_STEPS": "3", "DQ_TAIL_MAX": "0", "DQ_SMALL_N": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_MID_GROUPS": "256"},
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_SMALL_N": "0", "DQ_MID_GROUPS": "256", "DQ_PAIR_CHAINS": "0", "DQ_TAIL_MAX": "0"},  # ... many large groups, rank >> 8
]


@pytest.mark.parametrize("env", FUZZ_ENVS, ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "default")
def test_randomised_structured_inputs(ldss, oracle_mod, monkeypatch, env):
    """Differential test over ~400 structured inputs of 1 .. 2 000 000 bytes (sizes on both sides of every
    path switch: single-workgroup sorter, packed / pair keys, tie bits, sparse / dense finishing), under
    the default adaptive choices and with the alternatives forced."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(0xD17A + len(env))
    sizes = [int(x) for x in rng.integers(1, 2000, 120)] + [int(x) for x in rng.integers(2000, 20000, 120)] + \
            [int(x) for x in rng.integers(20000, 100000, 100)] + [int(x) for x in rng.integers(100000, 300000, 50)] + \
            [int(x) for x in rng.integers(300000, 2000000, 10)]
    for i, n in enumerate(sizes):
        T = structured_text(rng, n)
        ref = oracle_mod.divsufsort(T)
        if i % 9 == 0:
            assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), ref.astype(np.int64)), (env, i, n, "i64")
        else:
            assert np.array_equal(ldss.Sort(T), ref), (env, i, n)


@pytest.mark.parametrize("n", [1 << 16, (1 << 20) + 3, 5_000_000])
def test_uniform_random_matches_oracle(ldss, oracle_mod, n):
    T = oracle_mod.gen_uniform(n, 0x5EED0002)
    SA = ldss.Sort(T)
    assert oracle_mod.sufcheck(T, SA) == 0
    assert np.array_equal(SA, oracle_mod.divsufsort(T))


@pytest.mark.parametrize("n,R", [(300_000, 4096), (4_000_000, 64 * 1024)])
def test_enwik_like_text_matches_oracle(ldss, oracle_mod, n, R):
    T = oracle_mod.gen_enwik_like(n, 0xD17A0, R)
    SA = ldss.Sort(T)
    assert oracle_mod.sufcheck(T, SA) == 0
    assert np.array_equal(SA, oracle_mod.divsufsort(T))


def test_i64_entry_point_matches_i32(ldss, oracle_mod):
    for T in (oracle_mod.gen_enwik_like(200_000, 1, 4096), oracle_mod.gen_uniform(1_000_003, 5),
              load_asset("crash-04dc74e45e66386a3312a5a5825b020bcadc175c")):
        SA64 = ldss.Sort(T, index_dtype=np.int64)
        assert SA64.dtype == np.int64
        assert np.array_equal(SA64, oracle_mod.divsufsort(T).astype(np.int64))


def test_device_resident_entry_point(ldss, oracle_mod):
    import torch
    T = oracle_mod.gen_enwik_like(1_000_000, 3, 16384)
    dT = torch.from_numpy(T).cuda()
    dSA = ldss.Sort(dT)
    assert dSA.is_cuda and dSA.dtype == torch.int32
    assert np.array_equal(dSA.cpu().numpy(), oracle_mod.divsufsort(T))
    # unaligned device text (a slice) and caller-provided output
    dT2 = dT[3:777_777]
    out = torch.empty(dT2.numel(), dtype=torch.int32, device="cuda")
    ldss.Sort(dT2.contiguous(), out)
    assert np.array_equal(out.cpu().numpy(), oracle_mod.divsufsort(T[3:777_777]))


@pytest.mark.parametrize("copy_in_front", [False, True])
def test_device_text_is_copied_by_the_histogram_pass(ldss, oracle_mod, monkeypatch, copy_in_front):
    """The library's padded copy of a device-resident text is written by the pass that reads the text first
    (text_hist_kernel, 16-byte chunks + a byte tail) when the caller's buffer is 16-byte aligned, by a copy in front
    otherwise (DQ_TEXT_COPY=1 forces that): lengths around the chunking, offsets 0 .. 17 into an allocation, and a
    buffer that is followed by non-zero bytes (the pad behind the text must be the library's zeros, not the caller's)."""
    import torch
    if copy_in_front:
        monkeypatch.setenv("DQ_TEXT_COPY", "1")
    base = oracle_mod.gen_enwik_like(300_000, 9, 4096)
    big = torch.from_numpy(np.concatenate([base, np.full(4096, 0x41, np.uint8)])).cuda()      # 'A's behind every slice
    for off, n in [(0, 300_000), (0, 299_999), (0, 70_001), (16, 100_000), (16, 99_985), (1, 65_537), (17, 65_600), (32, 8_193), (0, 16_400)]:
        dT = big[off:off + n]
        got = ldss.Sort(dT).cpu().numpy()
        assert np.array_equal(got, oracle_mod.divsufsort(base[off:off + n])), (off, n, copy_in_front)


def test_full_size_config_by_properties(ldss, oracle_mod):
    """BASELINE configs[1]: 64 MiB uniform random, checked by the size-independent
    properties the reference's own Verify uses (sufcheck is O(n); sampled strict order)."""
    n = 64 << 20
    T = oracle_mod.gen_uniform(n, 0x5EED0002)
    SA = ldss.Sort(T)
    assert oracle_mod.sufcheck(T, SA) == 0
    assert oracle_mod.verify_sampled(T, SA, 1_000_000, 11) == -1


def test_batch_entry_point(backend_lib, oracle_mod):
    import ctypes
    # more inputs than pipeline slots (copy-in / sort / copy-out overlap, slots reused), of mixed sizes,
    # plus the ones that bypass the pipeline: empty, 1 and 2 bytes, short texts
    texts = [oracle_mod.gen_uniform(100_000 + 37_000 * (j % 5), 0x5EED0500 + j) for j in range(17)]
    texts.append(np.zeros(0, np.uint8))
    texts.append(oracle_mod.gen_enwik_like(50_000, 9, 4096))
    texts.append(np.array([7], np.uint8))
    texts.append(np.array([9, 3], np.uint8))
    texts.append(oracle_mod.net_random_bytes(4096))
    texts.append(oracle_mod.gen_enwik_like(700_000, 11, 16384))
    sas = [np.empty(t.size, np.int32) for t in texts]
    cnt = len(texts)
    tp = (ctypes.c_void_p * cnt)(*[t.ctypes.data if t.size else None for t in texts])
    sp = (ctypes.c_void_p * cnt)(*[s.ctypes.data if s.size else None for s in sas])
    ln = (ctypes.c_int64 * cnt)(*[t.size for t in texts])
    rc = backend_lib.dq_sufsort_hip_batch_i32(cnt, tp, ln, sp, 1, None)
    assert rc == 0, backend_lib.dq_last_error()
    for t, s in zip(texts, sas):
        assert np.array_equal(s, oracle_mod.divsufsort(t))


def test_batch_entry_point_two_device_slots(backend_lib, oracle_mod):
    """ndev > 1: the LPT split and the thread-per-device fan-out of dq_sufsort_hip_batch_i32.  A 1-GPU box runs it
    with devs = {0, 0}: two shares, two host threads, each with its own three-stage pipeline, taking turns on the
    one device (batch_mu); with more GPUs the same call also runs over all of them."""
    import ctypes
    sizes = [3_000_000, 40_000, 1_200_000, 9, 700_000, 2_500_000, 0, 300_000, 5000, 1_900_000, 650_000, 2, 810_000]
    texts = [oracle_mod.gen_uniform(s, 0x5EED0900 + j) if j % 3 else oracle_mod.gen_enwik_like(s, 77 + j, 8192)
             for j, s in enumerate(sizes)]
    cnt = len(texts)
    ln = (ctypes.c_int64 * cnt)(*[t.size for t in texts])
    tp = (ctypes.c_void_p * cnt)(*[t.ctypes.data if t.size else None for t in texts])
    expect = [oracle_mod.divsufsort(t) for t in texts]
    # (8 slots: the fan-out of a full node -- 8 shares, 8 host threads + 24 stage threads -- on whatever devices there are)
    ndevs = [(2, [0, 0]), (3, [0, 0, 0]), (8, [0] * 8)]
    have = backend_lib.dq_device_count()
    if 1 < have < 8:
        ndevs.append((8, [d % have for d in range(8)]))
    if have > 1:
        ndevs.append((have, list(range(have))))
    for ndev, devs in ndevs:
        sas = [np.full(t.size, -7, np.int32) for t in texts]
        sp = (ctypes.c_void_p * cnt)(*[s.ctypes.data if s.size else None for s in sas])
        dv = (ctypes.c_int32 * ndev)(*devs)
        rc = backend_lib.dq_sufsort_hip_batch_i32(cnt, tp, ln, sp, ndev, dv)
        assert rc == 0, backend_lib.dq_last_error()
        for j, (s, e) in enumerate(zip(sas, expect)):
            assert np.array_equal(s, e), (ndev, j)
        info = _abi_mod().last_batch_info()             # dq_last_batch_info: what the shares report back
        assert 0 <= info["pipelined"] <= cnt and info["slowest_share_ms"] > 0
        assert 0 <= info["shares_bound_to_numa_node"] <= ndev
        if info["pipelined"]:
            assert info["sort_ms"] > 0 and info["copy_in_ms"] > 0 and info["copy_out_ms"] > 0
    node = backend_lib.dq_device_numa_node(0)
    assert node >= -1 and backend_lib.dq_device_numa_node(10_000) == -1


def _abi_mod():
    from deltaq_amd import _abi
    return _abi


FORCED_PATHS = [
    {},                                                       # defaults (adaptive)
    {"DQ_BINNED_ISA": "1"},                                      # first inverse suffix array through suffix-binned words (default from 32 MiB on)
    {"DQ_BINNED_ISA": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_NO_FIRST_SMALL": "1"},
    {"DQ_PACKED": "1", "DQ_KEY_BYTES": "3", "DQ_SPARSE": "1"},   # packed words, finisher, sparse rounds
    {"DQ_PACKED": "1", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},   # packed words, dense doubling
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "3", "DQ_SPARSE": "1"},   # pairs, sparse + fallback to dense
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_SPARSE": "0"},   # pairs, 8-byte keys, dense doubling
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "1"},
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "4", "DQ_SPARSE": "0", "DQ_NO_SMALL": "1", "DQ_TAIL_MAX": "0"},   # doubling without the small-group rounds
    {"DQ_NO_FUSED_TIES": "1"},                                   # packed sort + general rebucket pass instead of tie bits
    {"DQ_PACKED": "1", "DQ_KEY_BYTES": "2"},                     # tie bits with MANY ties (dense doubling after them)
    {"DQ_FORCE_RSHIFT": "1"},                                    # doubling rounds with rank >> 1 in the composite key
    {"DQ_FORCE_RSHIFT": "1", "DQ_SPARSE": "1"},
    {"DQ_FORCE_RSHIFT": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_SPARSE": "0", "DQ_NO_FIRST_SMALL": "1", "DQ_NO_SMALL": "1", "DQ_BINNED_ISA": "1", "DQ_TAIL_MAX": "0"},   # ... on a list that came keyed from the binned first ISA
    {"DQ_CODED": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8"},    # coded round-0 keys (dq_alpha_code.h), dense doubling
    {"DQ_CODED": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_SPARSE": "1"},
    {"DQ_CODED": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "8", "DQ_NO_BINNED_ISA": "1", "DQ_NO_SMALL": "1", "DQ_TAIL_MAX": "0"},
    {"DQ_MID_GROUPS": "0"},                                      # the two-class rounds: small_group_round_kernel (<= 8 / 32) + radix passes
    {"DQ_MID_GROUPS": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_NO_CHAIN": "1", "DQ_TAIL_MAX": "0"},
    {"DQ_MID_GROUPS": "256", "DQ_TAIL_MAX": "0"},                                    # tie groups of up to 256 / 512 / 1024 members finished in LDS (dq_mid_groups.h; 1024 is the default)
    {"DQ_MID_GROUPS": "1024", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_TAIL_MAX": "0"},
    {"DQ_MID_GROUPS": "512", "DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_NO_CHAIN": "1"},
    {"DQ_RUNS": "1", "DQ_TAIL_MAX": "0"},                                            # runs of one byte ordered by their own structure (dq_runs.h), forced on
    {"DQ_RUNS": "1", "DQ_NO_SMALL": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_TAIL_MAX": "0"},   # ... through the radix rounds
    {"DQ_RUNS": "1", "DQ_MID_GROUPS": "256", "DQ_NO_CHAIN": "1", "DQ_PAIR_CHAINS": "0", "DQ_TAIL_MAX": "0"},
    {"DQ_RUNS": "0"},
    {"DQ_PAIR_CHAINS": "2", "DQ_TAIL_MAX": "0"},                                     # pair chains (dq_pair_chains.h) before / after every round
    {"DQ_PAIR_CHAINS": "2", "DQ_NO_BINNED_ISA": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_TAIL_MAX": "0"},
    {"DQ_PAIR_CHAINS": "1", "DQ_PACKED": "1", "DQ_KEY_BYTES": "2", "DQ_TAIL_MAX": "0"},
    {"DQ_NO_WIDE_SMALL": "1"},                                   # lists of more than n/2 entries through the radix rounds (the default before round 5)
    {"DQ_NO_L_SHIFT": "1", "DQ_TAIL_MAX": "0"},                                      # radix-list keys of the LDS-class rounds with the full rank (before round 5)
    {"DQ_UPD_WINDOW": "1", "DQ_TAIL_MAX": "0"},                                      # rank updates span by span in LDS, forced on every list
    {"DQ_UPD_WINDOW": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_BINNED_ISA": "1", "DQ_TAIL_MAX": "0"},
    {"DQ_UPD_WINDOW": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
    {"DQ_TAIL_MAX": "0"},                                        # no tail kernel: the device-wide rounds to the end (before round 5)
    {"DQ_TAIL_MAX": "100"},
    {"DQ_CHAIN_STEPS": "1"},                                     # one rank per member and round in the chains and the tail kernel
    {"DQ_CHAIN_STEPS": "1", "DQ_TAIL_MAX": "0", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
    {"DQ_CHAIN_STEPS": "3", "DQ_TAIL_MAX": "0", "DQ_MID_GROUPS": "256", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0"},
    {"DQ_TAIL_MAX": "4096", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_RUNS": "1"},
    {"DQ_NO_L_SHIFT": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_MID_GROUPS": "256", "DQ_TAIL_MAX": "0"},
    {"DQ_NO_WIDE_SMALL": "1", "DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_TAIL_MAX": "0"},
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "1", "DQ_SPARSE": "0", "DQ_BINNED_ISA": "1", "DQ_UPD_BIN_MIN": "1", "DQ_TAIL_MAX": "0"},   # wide lists, binned first ISA, binned updates
    {"DQ_PACKED": "0", "DQ_KEY_BYTES": "2", "DQ_SPARSE": "0", "DQ_RUNS": "1", "DQ_NO_UPD_WORDS": "1"},        # wide lists, run-order round, two-array updates
]


@pytest.mark.parametrize("env", FORCED_PATHS, ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "default")
def test_every_code_path_is_bit_exact(ldss, oracle_mod, monkeypatch, env):
    """The adaptive choices (key width, packed words, sparse/dense finishing) are forced in
    turn; every combination must produce the same (unique) suffix array."""
    monkeypatch.setenv("DQ_SMALL_N", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cases = [
        load_asset("crash-04dc74e45e66386a3312a5a5825b020bcadc175c"),
        load_asset("fuzz3"),
        oracle_mod.gen_uniform(3_000_000, 0x5EED0002),
        oracle_mod.gen_uniform(200_000, 11) & 3,
        oracle_mod.gen_enwik_like(500_000, 5, 16384),
        np.concatenate([oracle_mod.gen_uniform(100_000, 5), oracle_mod.gen_uniform(100_000, 5)[:30_000],
                        np.zeros(11, np.uint8)]),
        np.zeros(70_001, np.uint8),
        oracle_mod.net_random_bytes(8193),
    ]
    for T in cases:
        T = np.ascontiguousarray(T, dtype=np.uint8)
        SA = ldss.Sort(T)
        assert np.array_equal(SA, oracle_mod.divsufsort(T)), (env, T.size)
    T = oracle_mod.gen_uniform(1_000_003, 7)
    assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), oracle_mod.divsufsort(T).astype(np.int64))


@pytest.mark.parametrize("env", [{"DQ_UPD_WINDOW": "1"}, {"DQ_UPD_BIN": "1", "DQ_UPD_BIN_MIN": "1"}, {"DQ_UPD_BIN": "2", "DQ_UPD_BIN_MIN": "1"}],
                         ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()))
def test_update_filler_word_when_n_is_a_power_of_two(ldss, oracle_mod, monkeypatch, env):
    """The binned rank updates start on a 16-byte boundary: an all-ones filler word in front of an odd-placed list.  For
    n = 2^ib the filler's suffix field reads n - 1, a real suffix (round-5 advice): it must be skipped as a word, not by
    its suffix field.  Texts of 2^16 and 2^20 bytes whose end repeats an earlier stretch (so that suffix n - 1 - h is
    still tied when a round gathers ISA[n - 1]) with a genuine zero tail on some, both index widths (an int64 rank of
    2^(64 - ib) - 1 sorts last in its group; truncated to int32 it reads -1 and hides)."""
    monkeypatch.setenv("DQ_SMALL_N", "0")
    monkeypatch.setenv("DQ_TAIL_MAX", "0")
    monkeypatch.setenv("DQ_PACKED", "0")
    monkeypatch.setenv("DQ_SPARSE", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for n in (1 << 16, 1 << 20):
        for seed in range(6):
            rng = np.random.default_rng(1000 + seed)
            A = rng.integers(0, 3, 9000 + seed).astype(np.uint8)
            if seed % 2:
                A[-(3 + seed):] = 0                                 # genuine zeros against the zero padding
            B = (oracle_mod.gen_uniform(n - 2 * A.size, 40 + seed) & 3).astype(np.uint8)
            T = np.ascontiguousarray(np.concatenate([A, B, A]), dtype=np.uint8)
            assert T.size == n
            monkeypatch.setenv("DQ_KEY_BYTES", "2" if seed < 3 else "8")
            ref = oracle_mod.divsufsort(T)
            assert np.array_equal(ldss.Sort(T), ref), (env, n, seed, "int32")
            assert np.array_equal(ldss.Sort(T, index_dtype=np.int64), ref.astype(np.int64)), (env, n, seed, "int64")


@pytest.mark.parametrize("env", [{"DQ_SPLIT": "1"}, {"DQ_SPLIT": "1", "DQ_CODED": "1"}, {"DQ_SPLIT": "1", "DQ_CODED": "0"},
                                 {"DQ_SPLIT": "2"}, {"DQ_SPLIT": "2", "DQ_CODED": "0", "DQ_BINNED_ISA": "1"}, {"DQ_SPLIT": "0"}],
                         ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()))
def test_split_round0(ldss, oracle_mod, backend_lib, monkeypatch, env):
    """Round 0 as a sample sort (dq_split_round0.h; the default from 32 MiB of text-like input on, test_gpu_full_configs.py
    runs it at full size): forced on texts of 5 ... 9 MiB -- slots of 40 ... 72 entries, so every text has oversize buckets
    and the overflow list, its sort and the placement run; raw and coded keys; texts made of a few heavy keys, which the
    sorted sample gives away (DQ_SPLIT=1: left to the digit passes) or which fill the overflow list and fall back
    (DQ_SPLIT=2: the path is taken whatever the sample says); device and host entry points."""
    import ctypes
    import torch
    monkeypatch.setenv("DQ_PACKED", "0")
    monkeypatch.setenv("DQ_KEY_BYTES", "8")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rnd = oracle_mod.gen_uniform
    rep = rnd(1_200_000, 5)
    mode = env.get("DQ_SPLIT")
    # name -> (text, does the finish kernel run under DQ_SPLIT=1?  None: depends on what the sample says)
    cases = {
        "text 9 MiB": (oracle_mod.gen_enwik_like(9 << 20, 21, 65536), True),
        "text 5 MiB + 5": (oracle_mod.gen_enwik_like((5 << 20) + 5, 22, 16384), None),
        "uniform 6 MB": (rnd(6_000_000, 0x5EED0002), True),
        "16 symbols": (rnd(5_500_000, 8) & 15, True),
        "2 symbols (heavy keys)": (rnd(5_300_000, 9) & 1, False),
        "zeros (one key)": (np.zeros((5 << 20) + 1, np.uint8), False),
        "repeats + zero tail": (np.concatenate([rep, rnd(1_500_000, 6), rep[:400_000], rep, rnd(1_300_000, 7), np.zeros(13, np.uint8)]), True),
        "below the size the path takes": (oracle_mod.gen_enwik_like((5 << 20) - 1, 23, 16384), False),
    }

    def sort_counting_launches(T):
        backend_lib.dq_profile_reset()
        backend_lib.dq_profile_enable(1)
        sa = ldss.Sort(T)
        backend_lib.dq_profile_enable(0)
        n_l = ctypes.c_int64()
        backend_lib.dq_profile_get(21, ctypes.byref(n_l), None, None, None)          # DQ_K_SPLIT_FINISH
        return sa, n_l.value

    for name, (T, runs) in cases.items():
        T = np.ascontiguousarray(T, dtype=np.uint8)
        ref = oracle_mod.divsufsort(T)
        sa, launched = sort_counting_launches(T)
        assert np.array_equal(sa, ref), (env, name, "host")
        if mode == "0" or T.size < (5 << 20):
            assert launched == 0, (env, name, launched)
        elif mode == "2":
            assert launched == 1, (env, name, launched)                  # taken whatever the sample says
        elif runs is not None and not (name.startswith("text") and env.get("DQ_CODED") == "0"):
            # (raw 8-byte keys of a text: the sample may well call them too heavy -- either way is fine)
            assert launched == (1 if runs else 0), (env, name, launched)
        assert np.array_equal(ldss.Sort(torch.from_numpy(T).cuda()).cpu().numpy(), ref), (env, name, "device")


def run_heavy_texts(oracle_mod):
    """Texts made of runs: what dq_runs.h is for, and every edge of its run-length passes (16-byte segments,
    4096-byte chunks, runs that cross them, reach the end of the text, are followed by smaller / larger bytes)."""
    rng = np.random.default_rng(77)
    out = [np.zeros(70_001, np.uint8), np.full(1 << 20, 255, np.uint8), np.full(65_536, 7, np.uint8)]
    for total in (70_000, 300_000, 2_000_000):
        parts, size = [], 0
        while size < total:
            kind = int(rng.integers(0, 6))
            if kind <= 2:                                            # a run: lengths around the segment / chunk sizes and long ones
                ln = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 255, 4095, 4096, 4097, 8192, 12_289, 40_000, 100_003]))
                parts.append(np.full(ln, int(rng.choice([0, 0, 0, 1, 127, 255])), np.uint8))
            elif kind == 3:
                parts.append(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8))
            elif kind == 4:                                          # a few bytes between runs (what follows a run decides its place)
                parts.append(rng.integers(0, 3, int(rng.integers(1, 4)), dtype=np.uint8))
            else:                                                    # an earlier stretch again (runs inside repeats)
                if parts:
                    src = np.concatenate(parts[-8:])
                    a = int(rng.integers(0, src.size))
                    parts.append(src[a:a + int(rng.integers(1, 50_000))].copy())
            size = sum(p.size for p in parts)
        out.append(np.concatenate(parts)[:total])
    T = oracle_mod.gen_uniform(600_000, 3)
    T[100_000:233_000] = 0
    T[400_000:400_100] = 0
    T[-5000:] = 9                                                    # a run into the end of the text
    out.append(T)
    # a few long runs in text that is otherwise not made of runs (a shared library with a padding area: too little of
    # the text for the run lengths to be computed up front -- the rounds meet the runs' groups and switch to them late)
    T = oracle_mod.gen_enwik_like(3_000_000, 11, 65536)
    T[1_000_000:1_150_000] = 88
    T[2_000_000:2_040_000] = 88
    T[2_500_000:2_500_000 + 70_000] = np.resize(np.array([0, 255], np.uint8), 70_000)   # and a period-2 stretch, which stays with the rounds
    out.append(T)
    return out


@pytest.mark.parametrize("env", [{}, {"DQ_RUNS": "1"}, {"DQ_RUNS": "0"}, {"DQ_RUNS": "1", "DQ_NO_SMALL": "1"},
                                 {"DQ_RUNS": "1", "DQ_BINNED_ISA": "1"}, {"DQ_RUNS": "1", "DQ_BINNED_ISA": "1", "DQ_NO_FIRST_SMALL": "1"},
                                 {"DQ_RUNS": "1", "DQ_SPARSE": "1"}, {"DQ_RUNS": "1", "DQ_MID_GROUPS": "256", "DQ_NO_BINNED_ISA": "1"},
                                 {"DQ_LATE_RUNS_MIN": "1"}, {"DQ_LATE_RUNS_MIN": "64", "DQ_UPD_BIN_MIN": "1"}, {"DQ_NO_LATE_RUNS": "1"},
                                 {"DQ_LATE_RUNS_MIN": "1", "DQ_RUN_PERIOD": "1"}, {"DQ_LATE_RUNS_MIN": "1", "DQ_RUN_PERIOD": "2"},
                                 {"DQ_LATE_RUNS_MIN": "16", "DQ_RUN_PERIOD": "24"},
                                 {"DQ_RUNS": "1", "DQ_RUN_PERIOD": "5"}, {"DQ_RUNS": "1", "DQ_RUN_PERIOD": "3", "DQ_MID_GROUPS": "256"}],
                         ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "default")
def test_runs_of_one_byte(ldss, oracle_mod, monkeypatch, env):
    monkeypatch.setenv("DQ_SMALL_N", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for T in run_heavy_texts(oracle_mod):
        T = np.ascontiguousarray(T, dtype=np.uint8)
        assert np.array_equal(ldss.Sort(T), oracle_mod.divsufsort(T)), (env, T.size)


@pytest.mark.parametrize("env", [{}, {"DQ_MID_GROUPS": "0"}, {"DQ_SPARSE": "1"}, {"DQ_BINNED_ISA": "1"}, {"DQ_SPARSE": "0", "DQ_PAIR_CHAINS": "2"},
                                 {"DQ_NO_TWINS": "1"}, {"DQ_RUNS": "1", "DQ_RUN_PERIOD": "5"}, {"DQ_RUNS": "1", "DQ_RUN_PERIOD": "5", "DQ_NO_TWINS": "1"}],
                         ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "default")
def test_doubled_texts_finish_at_the_twin_pairs(ldss, oracle_mod, monkeypatch, env):
    """block + block, as the bzip2 encoder hands its blocks to the sorter (dq_bz2.h): suffix i + n/2 is a prefix of
    suffix i, the pair stays tied for n/2 - i characters; the sorter stops as soon as nothing but such pairs is left
    (twin_pairs_kernel) -- the suffix array is the same, entry for entry."""
    import torch
    monkeypatch.setenv("DQ_ASSUME_DOUBLED", "1")
    monkeypatch.setenv("DQ_SMALL_N", "0")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(31)
    zero_runs = np.tile(np.array([0, 0, 0, 0, 251], np.uint8), 40_000)             # a diff stream after its run-length pass
    zero_runs[rng.integers(0, zero_runs.size, 300)] = rng.integers(1, 256, 300).astype(np.uint8)
    blocks = [oracle_mod.gen_uniform(450_000, 3), oracle_mod.gen_enwik_like(300_000, 5, 8192), zero_runs,
              np.zeros(20_000, np.uint8), np.tile(np.frombuffer(b"ab", np.uint8), 9_000), np.tile(oracle_mod.gen_uniform(777, 2), 40),
              oracle_mod.gen_uniform(5, 1), oracle_mod.gen_uniform(4097, 9), oracle_mod.gen_uniform(3, 4).repeat(2000),
              rng.integers(0, 2, 100_000).astype(np.uint8), np.frombuffer(b"a", np.uint8)]
    for B in blocks:
        T = np.ascontiguousarray(np.concatenate([B, B]), dtype=np.uint8)
        ref = oracle_mod.divsufsort(T)
        assert np.array_equal(ldss.Sort(T), ref), (env, B.size, "host")
        assert np.array_equal(ldss.Sort(torch.from_numpy(T).cuda()).cpu().numpy(), ref), (env, B.size, "device")


def test_regression_inputs_found_by_the_stress_runs(ldss, oracle_mod, monkeypatch):
    """Inputs that tests/manual/stress.py caught a build on (tests/golden/regress/*.npy, each with the flags it ran under).
    stress_332_18050: suffix-binned first ISA on an input that leaves fewer than n/6 suffixes tied -- the finisher of the
    sparse path read a rank list that had been left in 32-bit form for the first LDS-class round (memory fault)."""
    import glob
    import os
    from conftest import GOLDEN_DIR
    cases = {"stress_332_18050_binned_isa_few_ties.npy": [{"DQ_BINNED_ISA": "1"}, {"DQ_BINNED_ISA": "1", "DQ_PAIR_CHAINS": "2"},
                                                          {"DQ_BINNED_ISA": "1", "DQ_SPARSE": "1"}, {}]}
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, "regress", "*.npy")))
    assert files
    for f in files:
        T = np.ascontiguousarray(np.load(f), dtype=np.uint8)
        ref = oracle_mod.divsufsort(T)
        for env in cases.get(os.path.basename(f), [{}]):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            import torch
            assert np.array_equal(ldss.Sort(torch.from_numpy(T).cuda()).cpu().numpy(), ref), (f, env, "device")
            assert np.array_equal(ldss.Sort(T), ref), (f, env, "host")
            for k in env:
                monkeypatch.delenv(k)


def test_shared_provider_from_many_threads(ldss, oracle_mod):
    """Providers are shared singletons in the reference's benchmark (SuffixSortingBenchmarks.cs:59-61):
    every entry point must be re-entrant.  8 threads, one shared instance, different inputs."""
    import threading
    inputs = [oracle_mod.gen_uniform(300_000 + 1111 * i, 100 + i) if i % 2 == 0
              else oracle_mod.gen_enwik_like(150_000 + 777 * i, 200 + i, 4096) for i in range(8)]
    outs = [None] * len(inputs)
    errs = []

    def work(i):
        try:
            for _ in range(3):
                outs[i] = ldss.Sort(inputs[i])
        except Exception as e:            # pragma: no cover
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(inputs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for T, SA in zip(inputs, outs):
        assert np.array_equal(SA, oracle_mod.divsufsort(T))


def test_shared_provider_overlaps_small_sorts(ldss, oracle_mod):
    """A device has several contexts (stream + workspace each) for texts of up to 4 MiB, so the threads of a host
    that shares one provider overlap their sorts instead of queueing behind one mutex: 8 threads x 4 KiB ... 1 MiB
    inputs get through faster together than one after the other (printed), and every result must be right (asserted)."""
    import threading
    import time
    sizes = [4096, 20_000, 65_536, 200_000, 300_000, 500_000, 800_000, 1 << 20]
    inputs = [oracle_mod.gen_enwik_like(n, 300 + i, 8192) if i % 2 else oracle_mod.gen_uniform(n, 400 + i) for i, n in enumerate(sizes)]
    expect = [oracle_mod.divsufsort(T) for T in inputs]
    reps = 12
    for T in inputs:
        ldss.Sort(T)                                         # warm: workspaces of every slot size class
    bad = []

    def work(i):
        for _ in range(reps):
            if not np.array_equal(ldss.Sort(inputs[i]), expect[i]):
                bad.append(i)

    t0 = time.perf_counter()
    for i in range(len(inputs)):
        work(i)
    serial = time.perf_counter() - t0
    best = None
    for _ in range(3):
        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(inputs))]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert not bad, bad
    # (the ratio is printed, not asserted: a wall-clock inequality on a shared box is a flake waiting for a noisy
    # neighbour, and under `pytest -x` it would hide every later test from the record -- round-4 verdict.  bench.py and
    # DESIGN.md section 3 carry the measured 2.5x.)
    print(f"8 inputs x {reps}: one thread {serial*1e3:.1f} ms, 8 threads {best*1e3:.1f} ms ({serial/best:.2f}x)")
