"""CPU oracle for the suffix-sorting hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product (``deltaq_amd``) never does.

It wraps ``oracle/libdq_oracle.so`` (plain C, built by ``oracle/Makefile``):
the restatement of the reference's LibDivSufSort (DivSufSort.cs / SsSort.cs /
TrSort.cs), the reference's own checkers (LibDivSufSortTests.Verify,
LDSSChecker.Check), a naive suffix array, the .NET ``Random`` generator used by
the reference's tests, and the synthetic workload generators.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdq_oracle.so")
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

CHECK_DONE = 0
CHECK_BAD_ARGUMENTS = -1
CHECK_OUT_OF_RANGE = -2
CHECK_WRONG_ORDER = -3
CHECK_WRONG_POSITION = -4


_MANIFEST = os.path.join(_HERE, "libdq_oracle.manifest")


def _source_digest() -> str:
    """sha256 over the names and contents of everything the library is built from."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(_HERE)):
        if f.endswith((".c", ".h")) or f == "Makefile":
            h.update(f.encode())
            with open(os.path.join(_HERE, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds)."""
    if force or _stale():
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:      # one builder at a time (bench.py's ranks)
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or _stale():
                # (-B: make orders by file times, which say nothing in a copied tree -- the decision is made by content)
                subprocess.run(["make", "-C", _HERE, "-s", "-B", "libdq_oracle.so"], check=True)
                with open(_MANIFEST + ".tmp", "w") as f:
                    f.write(_source_digest() + "\n")
                os.replace(_MANIFEST + ".tmp", _MANIFEST)
    return _LIB_PATH


def _stale() -> bool:
    """By content, like deltaq_amd/build.py (round-5 verdict): the library is current iff the manifest written beside it
    when it was built names the sources as they are now.  File times order nothing in a tree that was copied (the GPU
    box gets a snapshot): a checker built from other sources than the ones beside it must not vouch for anything."""
    if not os.path.exists(_LIB_PATH) or not os.path.exists(_MANIFEST):
        return True
    try:
        return open(_MANIFEST).read().strip() != _source_digest()
    except OSError:
        return True


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.c_void_p
        i64 = ctypes.c_int64
        for suf in ("i32", "i64"):
            getattr(L, f"dq_oracle_verify_strict_{suf}").restype = i64
            getattr(L, f"dq_oracle_verify_strict_{suf}").argtypes = [u8p, u8p, i64]
            getattr(L, f"dq_oracle_verify_sampled_{suf}").restype = i64
            getattr(L, f"dq_oracle_verify_sampled_{suf}").argtypes = [u8p, u8p, i64, i64, ctypes.c_uint64]
            getattr(L, f"dq_oracle_sufcheck_{suf}").restype = ctypes.c_int32
            getattr(L, f"dq_oracle_sufcheck_{suf}").argtypes = [u8p, i64, u8p, i64]
            getattr(L, f"dq_oracle_sufcheck_mt_{suf}").restype = ctypes.c_int32
            getattr(L, f"dq_oracle_sufcheck_mt_{suf}").argtypes = [u8p, i64, u8p, i64, ctypes.c_int32]
            getattr(L, f"dq_oracle_sais_{suf}").restype = ctypes.c_int32
            getattr(L, f"dq_oracle_sais_{suf}").argtypes = [u8p, u8p, i64]
            if hasattr(L, f"dq_oracle_divsufsort_{suf}"):
                getattr(L, f"dq_oracle_divsufsort_{suf}").restype = ctypes.c_int32
                getattr(L, f"dq_oracle_divsufsort_{suf}").argtypes = [u8p, u8p, i64]
        for suf in ("i32", "i64"):
            getattr(L, f"dq_oracle_bsdiff_search_{suf}").restype = ctypes.c_int32
            getattr(L, f"dq_oracle_bsdiff_search_{suf}").argtypes = [u8p, i64, u8p, u8p, i64, u8p, i64, i64, u8p, u8p]
            getattr(L, f"dq_oracle_bsdiff_scan_{suf}").restype = ctypes.c_int32
            getattr(L, f"dq_oracle_bsdiff_scan_{suf}").argtypes = [u8p, i64, u8p, u8p, i64] + [u8p] * 7
        L.dq_oracle_bspatch_apply.restype = ctypes.c_int32
        L.dq_oracle_bspatch_apply.argtypes = [u8p, i64, u8p, i64, u8p, i64, u8p, i64, i64, u8p]
        L.dq_oracle_naive_sa_i32.restype = ctypes.c_int32
        L.dq_oracle_naive_sa_i32.argtypes = [u8p, u8p, i64]
        L.dq_oracle_netrandom_bytes.restype = None
        L.dq_oracle_netrandom_bytes.argtypes = [ctypes.c_int32, u8p, i64]
        L.dq_oracle_netrandom_first_sample.restype = ctypes.c_int32
        L.dq_oracle_netrandom_first_sample.argtypes = [ctypes.c_int32]
        if hasattr(L, "dq_oracle_last_phase_seconds"):
            L.dq_oracle_last_phase_seconds.restype = None
            L.dq_oracle_last_phase_seconds.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def _text(text) -> np.ndarray:
    if isinstance(text, (bytes, bytearray, memoryview)):
        text = np.frombuffer(bytes(text), dtype=np.uint8)
    text = np.ascontiguousarray(text, dtype=np.uint8)
    return text


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data if a.size else 0


def _suf(sa: np.ndarray) -> str:
    if sa.dtype == np.int32:
        return "i32"
    if sa.dtype == np.int64:
        return "i64"
    raise TypeError(f"suffix array must be int32 or int64, not {sa.dtype}")


# ---- reference algorithm restated -------------------------------------------------
def divsufsort(text, dtype=np.int32) -> np.ndarray:
    """Restatement of LibDivSufSort.Sort(text) (LibDivSufSort.cs:12-19)."""
    T = _text(text)
    sa = np.empty(T.size, dtype=dtype)
    fn = getattr(lib(), f"dq_oracle_divsufsort_{_suf(sa)}")
    rc = fn(_ptr(T), _ptr(sa), T.size)
    if rc != 0:
        raise RuntimeError(f"oracle divsufsort failed: {rc}")
    return sa


def sais(text, dtype=np.int32) -> np.ndarray:
    """Restatement of SAIS.Sort(text) (SAIS.cs:14-41): the second, linear-time CPU implementation."""
    T = _text(text)
    sa = np.empty(T.size, dtype=dtype)
    rc = getattr(lib(), f"dq_oracle_sais_{_suf(sa)}")(_ptr(T), _ptr(sa), T.size)
    if rc != 0:
        raise RuntimeError(f"oracle sais failed: {rc}")
    return sa


def last_phase_seconds():
    out = (ctypes.c_double * 5)()
    lib().dq_oracle_last_phase_seconds(out)
    return dict(zip(("classify", "sssort", "trsort", "place_bstar", "induce"), list(out)))


def naive_sa(text) -> np.ndarray:
    T = _text(text)
    sa = np.empty(T.size, dtype=np.int32)
    lib().dq_oracle_naive_sa_i32(_ptr(T), _ptr(sa), T.size)
    return sa


# ---- reference checkers restated ---------------------------------------------------
def verify_strict(text, sa) -> int:
    """-1 when strictly sorted (LibDivSufSortTests.cs:43-59), else first bad i."""
    T = _text(text)
    sa = np.ascontiguousarray(sa)
    return int(getattr(lib(), f"dq_oracle_verify_strict_{_suf(sa)}")(_ptr(T), _ptr(sa), T.size))


def verify_sampled(text, sa, samples=1_000_000, seed=1) -> int:
    T = _text(text)
    sa = np.ascontiguousarray(sa)
    return int(getattr(lib(), f"dq_oracle_verify_sampled_{_suf(sa)}")(_ptr(T), _ptr(sa), T.size, samples, seed))


def sufcheck(text, sa) -> int:
    """LDSSChecker.Check result code (LDSSChecker.cs:23-119)."""
    T = _text(text)
    sa = np.ascontiguousarray(sa)
    return int(getattr(lib(), f"dq_oracle_sufcheck_{_suf(sa)}")(_ptr(T), T.size, _ptr(sa), sa.size))


def sufcheck_mt(text, sa, threads: int = 0) -> int:
    """LDSSChecker.Check evaluated by several threads (full-size configurations); same codes."""
    T = _text(text)
    sa = np.ascontiguousarray(sa)
    if threads <= 0:
        threads = min(64, os.cpu_count() or 1)
    return int(getattr(lib(), f"dq_oracle_sufcheck_mt_{_suf(sa)}")(_ptr(T), T.size, _ptr(sa), sa.size, threads))


def verify(text, sa) -> None:
    """LibDivSufSortTests.Verify: strict order, then sufcheck == Done."""
    rc = sufcheck(text, sa)  # range-check first so the strict loop cannot read out of bounds
    if rc == CHECK_OUT_OF_RANGE or rc == CHECK_BAD_ARGUMENTS:
        raise AssertionError(f"sufcheck returned {rc}")
    bad = verify_strict(text, sa)
    if bad >= 0:
        raise AssertionError(f"Input was unsorted at i={bad}, j={bad + 1}")
    if rc != CHECK_DONE:
        raise AssertionError(f"sufcheck returned {rc}")


# ---- the suffix array's consumer: Diff.Create's search and scan loop -------------------------------
def bsdiff_search(old, sa, new, scans=None, scan0=0, count=None):
    """Search(I, old, new[scan..], 0, n, out pos) (Diff.cs:267-298) for a batch of scan positions:
    returns (pos, len) arrays of sa's dtype."""
    O, N = _text(old), _text(new)
    sa = np.ascontiguousarray(sa)
    if scans is not None:
        scans = np.ascontiguousarray(scans, dtype=np.int64)
        count = scans.size
    elif count is None:
        count = N.size - scan0
    pos = np.empty(count, dtype=sa.dtype)
    ln = np.empty(count, dtype=sa.dtype)
    fn = getattr(lib(), f"dq_oracle_bsdiff_search_{_suf(sa)}")
    rc = fn(_ptr(O), O.size, _ptr(sa), _ptr(N), N.size, _ptr(scans) if scans is not None else 0, scan0, count,
            _ptr(pos), _ptr(ln))
    if rc != 0:
        raise RuntimeError(f"oracle bsdiff_search failed: {rc}")
    return pos, ln


def bsdiff_scan(old, sa, new):
    """The scan loop of Diff.Create (Diff.cs:91-232): returns (ctrl triples [k, 3] int64, diff bytes, extra
    bytes, number of Search calls)."""
    O, N = _text(old), _text(new)
    sa = np.ascontiguousarray(sa)
    m = N.size
    ctrl = np.empty(3 * (m + 1), dtype=np.int64)
    diff = np.empty(max(m, 1), dtype=np.uint8)
    extra = np.empty(max(m, 1), dtype=np.uint8)
    cnt = (ctypes.c_int64 * 4)()
    base = ctypes.addressof(cnt)
    fn = getattr(lib(), f"dq_oracle_bsdiff_scan_{_suf(sa)}")
    rc = fn(_ptr(O), O.size, _ptr(sa), _ptr(N), m, _ptr(ctrl), base, _ptr(diff), base + 8, _ptr(extra), base + 16,
            base + 24)
    if rc != 0:
        raise RuntimeError(f"oracle bsdiff_scan failed: {rc}")
    return ctrl[:3 * cnt[0]].reshape(-1, 3).copy(), diff[:cnt[1]].copy(), extra[:cnt[2]].copy(), int(cnt[3])


def bspatch_apply(old, ctrl, diff, extra, newsize) -> np.ndarray:
    """Patch.ApplyInternal (Patch.cs:95-168) on raw streams."""
    O = _text(old)
    ctrl = np.ascontiguousarray(ctrl, dtype=np.int64).reshape(-1)
    diff = np.ascontiguousarray(diff, dtype=np.uint8)
    extra = np.ascontiguousarray(extra, dtype=np.uint8)
    out = np.empty(newsize, dtype=np.uint8)
    rc = lib().dq_oracle_bspatch_apply(_ptr(O), O.size, _ptr(ctrl), ctrl.size // 3, _ptr(diff), diff.size,
                                       _ptr(extra), extra.size, newsize, _ptr(out))
    if rc != 0:
        raise RuntimeError(f"Corrupt patch ({rc})")
    return out


# ---- input generators ----------------------------------------------------------------
REFERENCE_SEED = 63 * 13 * 63 * 13  # LibDivSufSortTests.cs:29


def net_random_bytes(size: int, seed: int = REFERENCE_SEED) -> np.ndarray:
    """new Random(seed).NextBytes(new byte[size])."""
    out = np.empty(size, dtype=np.uint8)
    lib().dq_oracle_netrandom_bytes(seed, _ptr(out), size)
    return out


def net_random_first_sample(seed: int) -> int:
    return int(lib().dq_oracle_netrandom_first_sample(seed))


def gen_uniform(n: int, seed: int) -> np.ndarray:
    from tools import datagen
    return datagen.gen_uniform(n, seed)


def gen_enwik_like(n: int, seed: int = 0xD17A0, repeat_period: int = 256 * 1024) -> np.ndarray:
    from tools import datagen
    return datagen.gen_enwik_like(n, seed, repeat_period)
