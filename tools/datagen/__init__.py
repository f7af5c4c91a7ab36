"""Synthetic byte-text generators -- bench / test DATA only.

`gen_uniform`    i.i.d. uniform bytes, 8 little-endian bytes per splitmix64 draw
`gen_enwik_like` enwik8-style skewed text: Zipf word model, XML page boilerplate, injected
                 repeats up to 64 KiB every ~repeat_period bytes
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libdq_datagen.so")
_lib = None


_MANIFEST = os.path.join(_HERE, "libdq_datagen.manifest")


def _source_digest() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in ("gen.c", "Makefile"):
        with open(os.path.join(_HERE, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    """By content, not by file times (a copied tree's times order nothing): current iff the manifest beside the library
    names gen.c as it is now."""
    def stale() -> bool:
        return not os.path.exists(_LIB) or not os.path.exists(_MANIFEST) or open(_MANIFEST).read().strip() != _source_digest()
    if force or stale():
        import fcntl
        with open(os.path.join(_HERE, ".build.lock"), "w") as lock:      # one builder at a time (bench.py's ranks)
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                subprocess.run(["make", "-C", _HERE, "-s", "-B", "libdq_datagen.so"], check=True)
                with open(_MANIFEST + ".tmp", "w") as f:
                    f.write(_source_digest() + "\n")
                os.replace(_MANIFEST + ".tmp", _MANIFEST)
    return _LIB


def _load():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        L.dq_gen_uniform.restype = None
        L.dq_gen_uniform.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64]
        L.dq_gen_enwik_like.restype = None
        L.dq_gen_enwik_like.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64]
        _lib = L
    return _lib


def gen_uniform(n: int, seed: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    _load().dq_gen_uniform(out.ctypes.data if n else None, n, seed)
    return out


def gen_enwik_like(n: int, seed: int = 0xD17A0, repeat_period: int = 256 * 1024) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    _load().dq_gen_enwik_like(out.ctypes.data if n else None, n, seed, repeat_period)
    return out
