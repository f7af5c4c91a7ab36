"""Synthetic byte-text generators for benchmarks (BASELINE.json configs).

Pure numpy, integer-only, and bit-identical to the C generator the tests use
(tools/datagen/gen.c; checked by tests/test_workload.py).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_block(seed: int, count: int) -> np.ndarray:
    """First `count` outputs of splitmix64 seeded with `seed`."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + _GOLDEN * np.arange(1, count + 1, dtype=np.uint64)
        z = (x ^ (x >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def gen_uniform(n: int, seed: int) -> np.ndarray:
    """i.i.d. uniform bytes: 8 little-endian bytes per splitmix64 draw."""
    words = splitmix64_block(seed, (n + 7) // 8)
    return np.ascontiguousarray(words.astype("<u8").view(np.uint8)[:n])
