"""Random texts with the structure suffix sorters are sensitive to; one definition for the suite's fuzz test
(tests/test_gpu_parity.py), the long differential runs (tests/manual/stress.py) and tools/repro_fuzz.py."""
import numpy as np


def structured_text(rng, n):
    """Random text with the structure suffix sorters are sensitive to: a random alphabet size, runs of
    one byte, copies of earlier pieces (long repeats), periodic stretches and zero tails."""
    sigma = int(rng.choice([1, 2, 3, 4, 16, 64, 256]))
    out = []
    total = 0
    while total < n:
        kind = rng.integers(0, 6)
        ln = int(min(n - total, rng.integers(1, max(2, n // 3))))
        if kind == 0 or not out:
            piece = rng.integers(0, sigma, size=ln, dtype=np.uint8)
        elif kind == 1:
            piece = np.full(ln, rng.integers(0, sigma), dtype=np.uint8)
        elif kind == 2:                                         # copy of something earlier
            src = np.concatenate(out)
            a = int(rng.integers(0, src.size))
            piece = np.resize(src[a:a + ln], ln) if src[a:a + ln].size else src[:1]
        elif kind == 3:                                         # short period
            per = rng.integers(0, sigma, size=int(rng.integers(1, 9)), dtype=np.uint8)
            piece = np.resize(per, ln)
        elif kind == 4:
            piece = np.zeros(ln, dtype=np.uint8)
        else:
            piece = rng.integers(0, 256, size=ln, dtype=np.uint8)
        out.append(np.ascontiguousarray(piece, dtype=np.uint8))
        total += out[-1].size
    return np.concatenate(out)[:n]
