"""CPU tests: the oracle against the golden vectors and the reference's known answers."""
import hashlib

import numpy as np
import pytest

from conftest import asset_names, load_asset


def sha_i32(sa):
    return hashlib.sha256(np.asarray(sa).astype("<i4").tobytes()).hexdigest()


def test_net_random_check_values(oracle_mod, golden):
    # well-known first samples of .NET's seeded Random
    assert oracle_mod.net_random_first_sample(0) == 1559595546
    assert oracle_mod.net_random_first_sample(42) == 1434747710
    assert oracle_mod.net_random_bytes(16).tobytes().hex() == golden["net_random_first16_hex"]


def test_shruggy_known_answer(oracle_mod, golden):
    # LibDivSufSortTests.CheckShruggy (LibDivSufSortTests.cs:66-77)
    T = np.frombuffer("¯\\_(ツ)_/¯".encode("utf-8"), dtype=np.uint8)
    assert T.tolist() == golden["known_answers"]["shruggy"]["text"]
    want = [4, 8, 10, 2, 3, 9, 6, 7, 12, 1, 11, 0, 5]
    assert golden["known_answers"]["shruggy"]["sa"] == want
    assert oracle_mod.naive_sa(T).tolist() == want
    assert oracle_mod.divsufsort(T).tolist() == want


def test_readme_example(oracle_mod):
    # README.md:108
    assert oracle_mod.divsufsort(np.array([1, 2, 3, 4], dtype=np.uint8)).tolist() == [0, 1, 2, 3]


@pytest.mark.parametrize("name", asset_names())
def test_divsufsort_restatement_on_reference_fixtures(oracle_mod, golden, name):
    # LibDivSufSortTests.CheckFile (cs:108-124) + the two assets only the SAIS tests enumerate
    T = load_asset(name)
    g = golden["assets"][name]
    assert hashlib.sha256(T.tobytes()).hexdigest() == g["text_sha256"]
    sa = oracle_mod.divsufsort(T)
    oracle_mod.verify(T, sa)                       # the reference's own acceptance check
    assert sa[:8].tolist() == g["sa_head"]
    assert sha_i32(sa) == g["sa_sha256_le_i32"]


@pytest.mark.parametrize("size", [0, 1, 2, 4, 8, 16, 32, 51, 0x1000, 0x8000, 0x8000 - 1])
def test_divsufsort_restatement_on_reference_random_buffers(oracle_mod, golden, size):
    # LibDivSufSortTests.CheckRandomBuffer (cs:126-148)
    T = oracle_mod.net_random_bytes(size)
    g = golden["net_random_670761"][str(size)]
    assert hashlib.sha256(T.tobytes()).hexdigest() == g["text_sha256"]
    sa = oracle_mod.divsufsort(T)
    oracle_mod.verify(T, sa)
    assert sha_i32(sa) == g["sa_sha256_le_i32"]
    sa64 = oracle_mod.divsufsort(T, dtype=np.int64)
    assert np.array_equal(sa64, sa)


@pytest.mark.parametrize("name", asset_names())
def test_sais_restatement_on_reference_fixtures(oracle_mod, golden, name):
    # SAISTests enumerates every file of test/assets (the two LibDivSufSortTests skips included)
    T = load_asset(name)
    sa = oracle_mod.sais(T)
    oracle_mod.verify(T, sa)
    assert sha_i32(sa) == golden["assets"][name]["sa_sha256_le_i32"]
    assert np.array_equal(oracle_mod.sais(T, dtype=np.int64), sa)


@pytest.mark.parametrize("size", [0, 1, 2, 4, 8, 16, 32, 51, 0x1000, 0x8000, 0x8000 - 1])
def test_sais_restatement_on_reference_random_buffers(oracle_mod, golden, size):
    T = oracle_mod.net_random_bytes(size)
    sa = oracle_mod.sais(T)
    assert sha_i32(sa) == golden["net_random_670761"][str(size)]["sa_sha256_le_i32"]


def test_two_cpu_restatements_agree(oracle_mod):
    """SAIS (induced sorting) and LibDivSufSort (B* sorting + induction) restated independently: any
    translation slip in either shows up as a difference.  Known answers, random structured inputs, long
    repeats (deep SAIS recursion), medium sizes."""
    T = np.frombuffer("¯\\_(ツ)_/¯".encode("utf-8"), dtype=np.uint8)
    assert oracle_mod.sais(T).tolist() == [4, 8, 10, 2, 3, 9, 6, 7, 12, 1, 11, 0, 5]
    assert oracle_mod.sais(np.array([1, 2, 3, 4], dtype=np.uint8)).tolist() == [0, 1, 2, 3]
    rng = np.random.default_rng(5)
    for trial in range(200):
        n = int(rng.integers(1, 3000))
        sigma = int(rng.choice([1, 2, 3, 4, 16, 256]))
        T = rng.integers(0, sigma, n, dtype=np.uint8)
        if trial % 3 == 0:
            T[-min(n, 9):] = 0
        a = oracle_mod.sais(T)
        assert np.array_equal(a, oracle_mod.naive_sa(T)), (trial, n, sigma)
        assert np.array_equal(a, oracle_mod.divsufsort(T)), (trial, n, sigma)
    a, b = b"a", b"ab"
    while len(b) < 200_000:
        a, b = b, b + a
    cases = [np.frombuffer(b, dtype=np.uint8), np.zeros(100_000, np.uint8),
             np.tile(oracle_mod.gen_uniform(1000, 9), 300),
             oracle_mod.gen_uniform(1 << 21, 0x5EED0002), oracle_mod.gen_enwik_like(1 << 21, 0xD17A0, 64 * 1024)]
    for T in cases:
        sa = oracle_mod.sais(T)
        assert np.array_equal(sa, oracle_mod.divsufsort(T))
        assert oracle_mod.sufcheck(T, sa) == 0


def test_divsufsort_matches_naive_on_pathological_inputs(oracle_mod):
    cases = []
    for n in (3, 7, 8, 9, 63, 64, 65, 1023, 1024, 1025, 5000):
        cases += [np.zeros(n, np.uint8), np.full(n, 255, np.uint8),
                  np.tile(np.array([97, 98], np.uint8), n)[:n],
                  np.concatenate([np.full(n, 97, np.uint8), [98]]).astype(np.uint8),
                  np.concatenate([oracle_mod.net_random_bytes(n), np.zeros(9, np.uint8)])]
    a, b = b"a", b"ab"
    while len(b) < 6000:
        a, b = b, b + a
    cases.append(np.frombuffer(b, dtype=np.uint8))
    cases.append(np.array([bin(i).count("1") & 1 for i in range(4096)], dtype=np.uint8))
    cases.append(oracle_mod.gen_enwik_like(30000, 0xD17A0, 4096))
    cases.append(oracle_mod.gen_uniform(50000, 7) & 3)
    for T in cases:
        T = np.ascontiguousarray(T, dtype=np.uint8)
        sa = oracle_mod.divsufsort(T)
        assert np.array_equal(sa, oracle_mod.naive_sa(T))
        assert oracle_mod.sufcheck(T, sa) == 0


def test_divsufsort_medium_inputs_pass_reference_checkers(oracle_mod):
    for T in (oracle_mod.gen_uniform(1 << 20, 0x5EED0002),
              oracle_mod.gen_enwik_like(1 << 20, 0xD17A0, 64 * 1024),
              oracle_mod.net_random_bytes(1 << 18)):
        sa = oracle_mod.divsufsort(T)
        assert oracle_mod.sufcheck(T, sa) == 0
        assert oracle_mod.verify_sampled(T, sa, 200000, 3) == -1
        sa64 = oracle_mod.divsufsort(T, dtype=np.int64)
        assert np.array_equal(sa64, sa)


def test_checker_result_codes(oracle_mod):
    # LDSSChecker.ResultCode (LDSSChecker.cs:11-18)
    T = oracle_mod.net_random_bytes(100)
    sa = oracle_mod.naive_sa(T)
    assert oracle_mod.sufcheck(T, sa) == oracle_mod.CHECK_DONE
    assert oracle_mod.sufcheck(T, sa[:-1]) == oracle_mod.CHECK_BAD_ARGUMENTS
    bad = sa.copy(); bad[5] = 100
    assert oracle_mod.sufcheck(T, bad) == oracle_mod.CHECK_OUT_OF_RANGE
    bad = sa.copy(); bad[7] = -1
    assert oracle_mod.sufcheck(T, bad) == oracle_mod.CHECK_OUT_OF_RANGE
    bad = sa[::-1].copy()
    assert oracle_mod.sufcheck(T, bad) == oracle_mod.CHECK_WRONG_ORDER
    # same first characters, wrong suffix order -> WrongPosition
    Z = np.zeros(10, np.uint8)
    ident = np.arange(10, dtype=np.int32)
    assert oracle_mod.sufcheck(Z, ident) == oracle_mod.CHECK_WRONG_POSITION
    assert oracle_mod.verify_strict(Z, ident) == 0
    assert oracle_mod.verify_strict(Z, ident[::-1].copy()) == -1
    assert oracle_mod.sufcheck(np.zeros(0, np.uint8), np.zeros(0, np.int32)) == oracle_mod.CHECK_DONE


def test_threaded_checker_agrees_with_the_sequential_restatement(oracle_mod):
    """checkers_mt.c (used on the full-size configurations) returns the code of checkers.c's
    LDSSChecker.Check restatement on correct arrays and on corrupted ones, for any thread count."""
    rng = np.random.default_rng(7)
    for n in (1, 2, 5, 1000, 100_003):
        T = oracle_mod.gen_uniform(n, 5) & (3 if n % 2 else 255)
        sa = oracle_mod.divsufsort(T)
        for dt in (np.int32, np.int64):
            s = sa.astype(dt)
            for th in (1, 3, 8, 64):
                assert oracle_mod.sufcheck_mt(T, s, th) == oracle_mod.CHECK_DONE
            assert oracle_mod.sufcheck_mt(T, s[:-1], 4) == oracle_mod.CHECK_BAD_ARGUMENTS
            for trial in range(24 if n >= 5 else 0):
                b = s.copy()
                i, j = (int(x) for x in rng.integers(0, n, 2))
                kind = trial % 4
                if kind == 0:
                    b[i], b[j] = b[j], b[i]
                elif kind == 1:
                    b[i] = b[j]
                elif kind == 2:
                    b[i] = n + 5
                else:
                    b[i] = -1
                want = oracle_mod.sufcheck(T, b)
                for th in (1, 2, 7, 16):
                    assert oracle_mod.sufcheck_mt(T, b, th) == want, (n, trial, th)
    Z = np.zeros(1000, np.uint8)
    assert oracle_mod.sufcheck_mt(Z, np.arange(1000, dtype=np.int32), 5) == oracle_mod.CHECK_WRONG_POSITION


def test_prefix_doubling_model_matches_oracle(oracle_mod):
    """The numpy model of the GPU algorithm (tests/pd_model.py) against the oracle."""
    import pd_model
    for name in asset_names():
        T = load_asset(name)
        assert np.array_equal(pd_model.suffix_array(T), oracle_mod.divsufsort(T))
    for T in (np.zeros(65, np.uint8), np.array([5, 0, 0, 5, 0, 7, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 5, 0], np.uint8)):
        assert np.array_equal(pd_model.suffix_array(T), oracle_mod.naive_sa(T))
