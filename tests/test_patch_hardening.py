"""Hostile patches against the product's Patch.Apply (deltaq_amd/csrc/dq_bspatch.h), CPU only.

tests/native/patch_fuzz.cpp is built under AddressSanitizer + UndefinedBehaviorSanitizer and run as a program:
header lengths whose sum wraps, control triples that wrap the old-file position (the CVE-2014-9862 bug class),
decompression bombs in each stream, 4000 random mutations of a valid patch.  The reference answers all of these
with "Corrupt patch" (Patch.cs:131,140,153) or reads only newSize bytes (Patch.cs:115)."""
import ctypes
import os
import struct
import subprocess

import numpy as np

from conftest import ROOT

NATIVE = os.path.join(ROOT, "tests", "native")
CSRC = os.path.join(ROOT, "deltaq_amd", "csrc")


def test_hostile_patches_under_asan_ubsan():
    exe = os.path.join(NATIVE, "patch_fuzz")
    src = os.path.join(NATIVE, "patch_fuzz.cpp")
    deps = [src] + [os.path.join(CSRC, h) for h in ("dq_bspatch.h", "dq_bsdiff.h", "dq_bz2.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-pthread", src, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout


def test_host_pieces_from_six_threads_under_tsan():
    """dq_bz2.h / dq_bspatch.h / dq_alpha_code.h run on the callers' threads (three framing threads per Diff.Create):
    six threads at once under ThreadSanitizer -- no shared mutable state (the CRC table is a compile-time constant,
    the codeword tables are per call).  Two more threads each run the hand-over Diff.Create frames its growing streams
    with (producer appends and publishes a length, a follower feeds bz2::StreamEncoder, full blocks are encoded on
    threads of their own) and compare the result with the stream framed at once."""
    exe = os.path.join(NATIVE, "host_tsan")
    src = os.path.join(NATIVE, "host_tsan.cpp")
    deps = [src] + [os.path.join(CSRC, h) for h in ("dq_bspatch.h", "dq_bsdiff.h", "dq_bz2.h", "dq_alpha_code.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", src, "-o", exe, "-pthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if "unexpected memory mapping" in r.stderr:                    # (a kernel whose ASLR layout TSan cannot shadow)
        import pytest
        pytest.skip("ThreadSanitizer cannot run on this kernel")
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    assert "ThreadSanitizer" not in r.stderr, r.stderr


def packed(y):
    b = bytearray(struct.pack("<Q", abs(y)))
    if y < 0:
        b[7] |= 0x80
    return bytes(b)


def test_c_abi_rejects_the_two_advisor_patches(backend_lib):
    """The two proofs of concept of ADVICE round 2 through the product's C ABI (dq_bspatch_apply is host code)."""
    import bz2
    L = backend_lib
    old = np.arange(64, dtype=np.uint8)
    out = np.zeros(64, np.uint8)
    out_len = ctypes.c_int64(0)

    def run(patch):
        return L.dq_bspatch_apply(old.ctypes.data, old.size, patch, len(patch), out.ctypes.data, out.size, ctypes.byref(out_len))

    p1 = b"BSDIFF40" + packed(1 << 62) + packed(1 << 62) + packed(8) + b"\0" * 64
    assert run(p1) == -1 and b"Corrupt patch" in L.dq_last_error()
    ctrl = b"".join(packed(v) for v in (0, 0, 2**63 - 1, 1, 0, 0))
    zc, zd, ze = bz2.compress(ctrl), bz2.compress(b"\0" * 8), bz2.compress(b"")
    p2 = b"BSDIFF40" + packed(len(zc)) + packed(len(zd)) + packed(1) + zc + zd + ze
    assert run(p2) == -1 and b"Corrupt patch" in L.dq_last_error()
    # and a bomb: 64 MiB of zeros in the diff stream of a 4-byte file is cut at 4 bytes
    zd = bz2.compress(b"\0" * (64 << 20))
    zc = bz2.compress(b"".join(packed(v) for v in (4, 0, 0)))
    p3 = b"BSDIFF40" + packed(len(zc)) + packed(len(zd)) + packed(4) + zc + zd + ze
    assert run(p3) == 0 and out_len.value == 4 and out[:4].tolist() == [0, 1, 2, 3]
