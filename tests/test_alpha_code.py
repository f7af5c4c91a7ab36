"""CPU tests of the coded round 0 (deltaq_amd/csrc/dq_alpha_code.h, dq_coded_keys.h): the code is an alphabetic
prefix code with lengths in [4, 8] and minimal expected length among those; the 64-bit keys the device builds from
it are the first 64 bits of the concatenated codewords and order suffixes consistently with the suffix array."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

NATIVE = os.path.join(ROOT, "tests", "native")


@pytest.fixture(scope="module")
def alpha():
    so, src = os.path.join(NATIVE, "libalpha_harness.so"), os.path.join(NATIVE, "alpha_harness.cpp")
    hdrs = [os.path.join(ROOT, "deltaq_amd", "csrc", h) for h in ("dq_alpha_code.h", "dq_coded_keys.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in [src] + hdrs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.t_alpha_code.restype = ctypes.c_int
    L.t_alpha_code.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.t_coded_keys.restype = None
    L.t_coded_keys.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]

    class A:
        @staticmethod
        def code(hist):
            hist = np.ascontiguousarray(hist, np.int64)
            tab = np.zeros(256, np.uint16)
            avg = ctypes.c_double()
            sigma = L.t_alpha_code(hist.ctypes.data, tab.ctypes.data, ctypes.byref(avg))
            return sigma, tab, avg.value

        @staticmethod
        def keys(text, tab):
            buf = np.concatenate([np.ascontiguousarray(text, np.uint8), np.zeros(32, np.uint8)])
            out = np.zeros(max(text.size, 1), np.uint64)
            L.t_coded_keys(buf.ctypes.data, text.size, tab.ctypes.data, out.ctypes.data)
            return out[:text.size]
    return A


def optimal_cost(w, lmin=4, lmax=8):
    """Plain cubic dynamic programme (no bounded split search): the reference optimum."""
    s = len(w)
    P = np.concatenate([[0], np.cumsum(w)]).astype(np.float64)
    inf = float("inf")
    prev = None
    for r in range(0, lmax + 1):
        depth = lmax - r
        cur = np.full((s + 1, s + 1), inf)
        for i in range(s):
            cur[i][i + 1] = max(0, lmin - depth) * w[i]
        if r > 0:
            for ln in range(2, s + 1):
                for i in range(0, s - ln + 1):
                    j = i + ln
                    ks = np.arange(i + 1, j)
                    c = prev[i, ks] + prev[ks, j]
                    cur[i][j] = c.min() + P[j] - P[i]
        prev = cur
    return prev[0][s]


def histograms():
    rng = np.random.default_rng(3)
    out = []
    for sigma in (1, 2, 3, 15, 16, 17, 40, 73, 128, 129, 200, 256):
        for shape in range(3):
            h = np.zeros(256, np.int64)
            syms = np.sort(rng.choice(256, sigma, replace=False))
            if shape == 0:
                h[syms] = rng.integers(1, 1000, sigma)
            elif shape == 1:
                h[syms] = np.maximum(1, (1e7 / (1 + rng.permutation(sigma)) ** 1.3).astype(np.int64))      # Zipf
            else:
                h[syms] = 1
                h[syms[rng.integers(0, sigma)]] = 10 ** 9                                                   # one dominant byte
            out.append(h)
    return out


def test_code_is_alphabetic_prefix_free_and_length_limited(alpha):
    for h in histograms():
        sigma, tab, avg = alpha.code(h)
        assert sigma == int((h > 0).sum())
        present = np.flatnonzero(h)
        lens = (tab & 15).astype(int)
        codes = (tab >> 4).astype(int)
        assert np.all(lens[h == 0] == 8) and np.all(codes[h == 0] == 0)
        assert np.all((lens[present] >= 4) & (lens[present] <= 8))
        assert codes[present[0]] == 0                                            # smallest byte: all zeros
        words = ["{:0{}b}".format(codes[b], lens[b]) for b in present]
        for a, b in zip(words, words[1:]):
            assert a < b and not b.startswith(a) and not a.startswith(b)         # alphabetic + prefix free (sorted => all pairs)
        assert sum(2.0 ** -len(x) for x in words) <= 1.0 + 1e-12
        assert abs(avg - float((h[present] * lens[present]).sum()) / float(h.sum())) < 1e-9


def test_code_is_optimal_for_its_constraints(alpha):
    rng = np.random.default_rng(9)
    for trial in range(25):
        sigma = int(rng.integers(2, 60))
        h = np.zeros(256, np.int64)
        syms = np.sort(rng.choice(256, sigma, replace=False))
        h[syms] = np.maximum(1, (rng.random(sigma) ** 4 * 1e6).astype(np.int64))
        _, tab, avg = alpha.code(h)
        want = optimal_cost(h[syms].astype(np.float64))
        got = float((h[syms] * (tab[syms] & 15)).sum())
        assert got == want, (trial, sigma, got, want)


def spec_keys(text, tab):
    """first 64 bits of the concatenated codewords from every position, zeros behind the end"""
    lens = (tab & 15).astype(np.int64)
    codes = (tab >> 4).astype(np.uint64)
    n = text.size
    out = np.zeros(n, np.uint64)
    for i in range(n):
        acc, bits, j = 0, 0, i
        while bits < 64 and j < n:
            acc = (acc << int(lens[text[j]])) | int(codes[text[j]])
            bits += int(lens[text[j]])
            j += 1
        acc = acc >> (bits - 64) if bits >= 64 else acc << (64 - bits)
        out[i] = acc
    return out


def test_keys_match_the_specification_and_order_like_the_suffix_array(alpha, oracle_mod):
    rng = np.random.default_rng(17)
    texts = [oracle_mod.gen_enwik_like(3000, 5, 512), rng.integers(0, 4, 2000, dtype=np.uint8),
             (rng.integers(0, 256, 2500) * (rng.integers(0, 3, 2500) == 0)).astype(np.uint8),
             np.frombuffer(b"abracadabra" * 50 + b"\x00\x00\x00abra", dtype=np.uint8),
             rng.integers(0, 256, 1001, dtype=np.uint8), np.zeros(777, np.uint8), np.frombuffer(b"z", dtype=np.uint8)]
    for t in texts:
        t = np.ascontiguousarray(t)
        _, tab, _ = alpha.code(np.bincount(t, minlength=256))
        keys = alpha.keys(t, tab)
        assert np.array_equal(keys, spec_keys(t, tab))
        sa = oracle_mod.divsufsort(t).astype(np.int64)
        ks = keys[sa]
        assert np.all(ks[1:] >= ks[:-1])                                         # monotone along the suffix array
        tied = np.flatnonzero(ks[1:] == ks[:-1])
        b = t.tobytes()
        for p in tied[:: max(1, tied.size // 300)]:                              # equal keys: >= 8 equal characters (or the end)
            a, c = int(sa[p]), int(sa[p + 1])
            assert b[a:a + 8] == b[c:c + 8][: len(b[a:a + 8])] or b[c:c + 8] == b[a:a + 8][: len(b[c:c + 8])]


def test_doubling_from_coded_keys_gives_the_suffix_array(alpha, oracle_mod):
    """The whole pipeline in the numpy model (tests/pd_model.py) with the coded keys as round 0 and h = 8 from there
    on, dense and sparse finishing: the suffix array of the oracle."""
    import pd_model
    rng = np.random.default_rng(3)
    texts = [oracle_mod.gen_enwik_like(20_000, 7, 2048), rng.integers(0, 3, 5000, dtype=np.uint8),
             np.concatenate([rng.integers(0, 200, 3000, dtype=np.uint8), np.zeros(40, np.uint8)]),
             np.frombuffer(b"mississippi" * 300, dtype=np.uint8), np.zeros(3000, np.uint8),
             np.tile(rng.integers(0, 9, 37, dtype=np.uint8), 120)]
    for t in texts:
        t = np.ascontiguousarray(t)
        _, tab, _ = alpha.code(np.bincount(t, minlength=256))
        keys = alpha.keys(t, tab)
        want = oracle_mod.divsufsort(t).astype(np.int64)
        for sparse in (False, True):
            got = pd_model.suffix_array(t, kbytes=8, sparse=sparse, keys=keys)
            assert np.array_equal(got, want), (t.size, sparse)
