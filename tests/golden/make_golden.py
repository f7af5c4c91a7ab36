#!/usr/bin/env python3
"""Regenerates tests/golden/golden.json.

Golden vectors for the suffix-sorting path.  The reference (C#) cannot run in this image,
and its own tests store no expected arrays (they are property tests: strict order +
sufcheck, LibDivSufSortTests.cs:43-64), so the vectors are derived from the *contract*:
the suffix array under SequenceCompareTo order is unique.  Each entry is produced by the
naive comparison sort in oracle/checkers.c and accepted by the restated reference checkers
(Verify + LDSSChecker.Check) before its digest is written.

Inputs covered:
  * every file in the reference's test/assets (copied byte-for-byte to tests/golden/assets)
  * new Random(670761).NextBytes(size) for the sizes of CheckRandomBuffer
    (LibDivSufSortTests.cs:126-137)
  * the known-answer strings (CheckShruggy, LibDivSufSortTests.cs:66-77; README.md:108)

Run from the repo root:  python tests/golden/make_golden.py
"""
import glob
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

RANDOM_SIZES = [0, 1, 2, 4, 8, 16, 32, 51, 0x1000, 0x8000 - 1, 0x8000]


def entry(T: np.ndarray) -> dict:
    sa = oracle.naive_sa(T)
    oracle.verify(T, sa)
    return {
        "n": int(T.size),
        "text_sha256": hashlib.sha256(T.tobytes()).hexdigest(),
        "sa_head": sa[:8].tolist(),
        "sa_sha256_le_i32": hashlib.sha256(sa.astype("<i4").tobytes()).hexdigest(),
    }


def main() -> None:
    out = {"assets": {}, "net_random_670761": {}, "known_answers": {}}
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "assets", "*"))):
        out["assets"][os.path.basename(path)] = entry(np.fromfile(path, dtype=np.uint8))
    stream = oracle.net_random_bytes(max(RANDOM_SIZES))
    out["net_random_first16_hex"] = stream[:16].tobytes().hex()
    for size in RANDOM_SIZES:
        T = oracle.net_random_bytes(size)
        assert np.array_equal(T, stream[:size])       # every size is a prefix of one stream
        out["net_random_670761"][str(size)] = entry(T)
    shruggy = np.frombuffer("¯\\_(ツ)_/¯".encode("utf-8"), dtype=np.uint8)
    e = entry(shruggy)
    e["text"] = shruggy.tolist()
    e["sa"] = oracle.naive_sa(shruggy).tolist()
    out["known_answers"]["shruggy"] = e
    readme = np.array([1, 2, 3, 4], dtype=np.uint8)
    e = entry(readme)
    e["text"] = readme.tolist()
    e["sa"] = oracle.naive_sa(readme).tolist()
    out["known_answers"]["readme_1234"] = e
    with open(os.path.join(ROOT, "tests", "golden", "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
