"""CPU tests of the host-side half of the BSDIFF40 row: the bzip2 codec (deltaq_amd/csrc/dq_bz2.h) against libbz2
(Python's bz2) in both directions, packed longs / header / Patch.Apply against patches framed by libbz2 from the
oracle's raw streams.  The encoder's block transform is the GPU sorter's job in the product; here a naive
sorter stands in (tests/native/bz2_harness.cpp), so that the format logic is checked without a GPU."""
import bz2
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

NATIVE = os.path.join(ROOT, "tests", "native")


@pytest.fixture(scope="module")
def harness():
    so = os.path.join(NATIVE, "libbz2_harness.so")
    src = os.path.join(NATIVE, "bz2_harness.cpp")
    hdr = os.path.join(ROOT, "deltaq_amd", "csrc", "dq_bz2.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", so], check=True)
    L = ctypes.CDLL(so)
    L.t_bz2_compress.restype = ctypes.c_int64
    L.t_bz2_compress.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
    L.t_bz2_decompress.restype = ctypes.c_int64
    L.t_bz2_decompress.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]

    class H:
        @staticmethod
        def compress(b, level=9):
            out = np.empty(len(b) * 2 + 1000, np.uint8)
            r = L.t_bz2_compress(b, len(b), out.ctypes.data, out.size, level)
            assert r >= 0, r
            return out[:r].tobytes()

        @staticmethod
        def decompress(b, cap):
            out = np.empty(cap + 16, np.uint8)
            r = L.t_bz2_decompress(b, len(b), out.ctypes.data, out.size)
            return r, out[:max(r, 0)].tobytes()
    return H


def sample_inputs(oracle_mod):
    rng = np.random.default_rng(1)
    cases = [b"", b"a", b"ab", b"aaaa", b"aaaaa", b"a" * 255, b"a" * 256, b"a" * 259, b"a" * 1000, b"abc" * 1000,
             bytes(range(256)) * 10, b"banana", b"\x00" * 70000]
    for n in (10, 100, 1000, 5000, 30000):
        cases.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        cases.append(rng.integers(0, 3, n, dtype=np.uint8).tobytes())
        cases.append((rng.integers(0, 256, n, dtype=np.uint8) * (rng.integers(0, 10, n) == 0)).astype(np.uint8).tobytes())
    cases.append(oracle_mod.gen_enwik_like(120_000, 3, 4096).tobytes())
    return cases


def test_stream_fed_in_pieces_is_the_stream_fed_at_once(harness, oracle_mod):
    """StreamEncoder.feed() behind a producer (Diff.Create frames the diff / extra streams while the scan still runs):
    the same bytes as one sweep over the finished stream, wherever the pieces end -- inside runs, on block boundaries,
    empty pieces -- and with full blocks encoded on their own threads meanwhile."""
    L = ctypes.CDLL(os.path.join(NATIVE, "libbz2_harness.so"))
    L.t_bz2_compress_fed.restype = ctypes.c_int64
    L.t_bz2_compress_fed.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                     ctypes.c_int64, ctypes.c_uint32]
    rng = np.random.default_rng(8)
    runs = np.repeat(rng.integers(0, 4, 3000, dtype=np.uint8), rng.integers(1, 700, 3000)).tobytes()      # runs across pieces
    cases = sample_inputs(oracle_mod) + [runs[:250_000], rng.integers(0, 256, 320_000, dtype=np.uint8).tobytes(),
                                         b"\x00" * 99_981 + b"\x01" * 99_981, b"ab" * 49_990 + b"b"]
    for k, c in enumerate(cases):
        level = 1 if len(c) > 60_000 else 9                  # (several blocks without megabytes for the naive sorter)
        want = harness.compress(c, level)
        for step, seed in ((1, 1), (7, 2), (300, 3), (5000, 4), (200_000, 5)):
            if step == 1 and len(c) > 5000:
                continue
            out = np.empty(len(c) * 2 + 1000, np.uint8)
            r = L.t_bz2_compress_fed(c, len(c), out.ctypes.data, out.size, level, step, seed)
            assert r >= 0 and out[:r].tobytes() == want, (k, len(c), step)
        assert bz2.decompress(want) == c


_REV8 = bytes(int(f"{i:08b}"[::-1], 2) for i in range(256))


def crc32_bzip2(data):
    """bzip2's CRC (polynomial 0x04c11db7, most significant bit first) is zlib's with every byte and the result mirrored."""
    import zlib
    return int(f"{zlib.crc32(bytes(data).translate(_REV8)) & 0xffffffff:032b}"[::-1], 2)


def rle1_blocks(data, level, span=0):
    """bzip2's first run-length pass and its block cut, written out plainly: runs of 4..255 equal bytes become four bytes
    and a count; a block closes when fewer than 5 bytes of room are left in its level * 100000 - 19 -- or, with span > 0
    (the product: dq::bz2::kBlockSpan), once it covers span bytes of the stream."""
    a = np.frombuffer(data, np.uint8)
    block_max = level * 100000 - 19
    starts = np.flatnonzero(np.concatenate([[True], a[1:] != a[:-1]])) if a.size else np.zeros(0, np.int64)
    ends = np.concatenate([starts[1:], [a.size]]) if a.size else starts
    blocks, spans, cur, cur_start, pos = [], [], bytearray(), 0, 0
    for s0, e0 in zip(starts.tolist(), ends.tolist()):
        c, left = int(a[s0]), e0 - s0
        while left > 0:
            if len(cur) + 5 > block_max or (span and pos - cur_start >= span):
                blocks.append(bytes(cur)); spans.append((cur_start, pos)); cur, cur_start = bytearray(), pos
            run = min(left, 255)
            cur += bytes([c]) * 4 + bytes([run - 4]) if run >= 4 else bytes([c]) * run
            left -= run; pos += run
    if a.size:
        blocks.append(bytes(cur)); spans.append((cur_start, pos))
    return blocks, spans


def test_prepass_cuts_the_blocks_of_the_definition(harness, oracle_mod):
    """Long stretches of one byte take a short cut through the pre-pass (runs of 255 counted 8 bytes a step, written as
    their five bytes) and through the CRC (zero bytes stepped over with the zero-byte operator): the blocks, their
    boundaries and their CRCs are those of the plain definition -- around multiples of 255, across block boundaries
    (level 1: a block is 99 981 coded bytes = 5 MB of zeros), wherever the pieces fed end."""
    L = ctypes.CDLL(os.path.join(NATIVE, "libbz2_harness.so"))
    L.t_bz2_prepass.restype = ctypes.c_int64
    L.t_bz2_prepass.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p,
                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.t_crc_bitwise.restype = ctypes.c_uint32
    L.t_crc_bitwise.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    L.t_crc_sliced.restype = ctypes.c_uint32
    L.t_crc_sliced.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    rng = np.random.default_rng(12)
    cases = [b"\x00" * k for k in (254, 255, 256, 509, 510, 511, 764, 765, 766, 1020, 255 * 40, 255 * 40 + 3)]
    cases += [b"\x07" * 5_200_000 + b"xyz" + b"\x00" * 6_000_000,                  # block boundaries inside the stretches
              b"\x00" * 5_099_031 + b"ab" + b"\x00" * 300, b"\x00" * 5_099_030 + b"ab", b"\x00" * 5_098_776 + b"\x01" * 600]
    mixed = bytearray()
    for _ in range(300):                                    # stretches of every length between noise
        mixed += bytes([int(rng.integers(0, 3))]) * int(rng.integers(1, 3000)) + rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8).tobytes()
    cases.append(bytes(mixed))
    L.t_bz2_prepass_span.restype = ctypes.c_int64
    L.t_bz2_prepass_span.argtypes = L.t_bz2_prepass.argtypes + [ctypes.c_int64]
    L.t_bz2_block_span.restype = ctypes.c_int64
    assert L.t_bz2_block_span() == 4 << 20
    # (the plain definition; then blocks that also end once they cover `span` bytes of the stream -- the product's 4 MiB,
    # and spans that fall inside runs, on multiples of 255 and between the pieces fed)
    for k, c in enumerate(cases):
        for span in (0, 4 << 20, 255 * 1000, 100_003, 1 << 20):
            if span not in (0, 4 << 20) and len(c) < 100_000:
                continue
            want, spans = rle1_blocks(c, 1, span)
            for step, seed in ((10_000_000, 1), (1000, 2), (257, 3), (70_000, 4)):
                rle = np.empty(len(c) + 64, np.uint8); lens = np.zeros(4096, np.uint32); crcs = np.zeros(4096, np.uint32)
                nb = L.t_bz2_prepass_span(c, len(c), 1, step, seed, rle.ctypes.data, rle.size, lens.ctypes.data, crcs.ctypes.data, 4096, span)
                assert nb == len(want), (k, span, step, nb, len(want))
                off = 0
                for b in range(nb):
                    assert rle[off:off + lens[b]].tobytes() == want[b], (k, span, step, b)
                    off += int(lens[b])
                    piece = c[spans[b][0]:spans[b][1]]
                    assert int(crcs[b]) == crc32_bzip2(piece), (k, span, step, b)
    assert crc32_bzip2(b"123456789") == 0xfc891918 == L.t_crc_bitwise(b"123456789", 9)      # (the catalogue's check value)
    # the stepped-over zero bytes against the bit-by-bit definition
    for n in (63, 64, 65, 71, 72, 1000, 4096 + 3, 100_000):
        for lead in (b"", b"\x01", b"abc", b"\x00\x00\x05"):
            buf = lead + b"\x00" * n + b"\x09" + b"\x00" * (n // 2)
            assert L.t_crc_sliced(buf, len(buf)) == L.t_crc_bitwise(buf, len(buf)), (n, lead)


def test_crc_variants_agree_with_the_definition(harness, oracle_mod):
    so = os.path.join(NATIVE, "libbz2_harness.so")
    L = ctypes.CDLL(so)
    for f in (L.t_crc_bitwise, L.t_crc_sliced, L.t_crc_mt):
        f.restype = ctypes.c_uint32
        f.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    rng = np.random.default_rng(4)
    for n in (0, 1, 7, 8, 9, 63, 1000, (1 << 20) + 5, (2 << 20) - 1, 2 << 20, (5 << 20) + 123):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        want = L.t_crc_bitwise(b, n) if n <= (1 << 20) + 5 else L.t_crc_sliced(b, n)
        assert L.t_crc_sliced(b, n) == want and L.t_crc_mt(b, n) == want, n
    z = bytes(9 << 20)                                       # a run of zeros with one byte set, split over four threads
    z2 = z[:5_000_000] + b"\x01" + z[5_000_001:]
    assert L.t_crc_mt(z2, len(z2)) == L.t_crc_sliced(z2, len(z2)) != L.t_crc_mt(z, len(z))


def test_decoder_reads_libbz2_streams(harness, oracle_mod):
    for i, c in enumerate(sample_inputs(oracle_mod)):
        for level in (1, 9):
            r, d = harness.decompress(bz2.compress(c, level), len(c))
            assert r == len(c) and d == c, (i, level)
    r, d = harness.decompress(bz2.compress(b"hello ") + bz2.compress(b"world"), 100)       # concatenated streams
    assert d == b"hello world"
    multi = oracle_mod.gen_enwik_like(450_000, 3, 4096).tobytes()                          # several 100 kB blocks
    assert harness.decompress(bz2.compress(multi, 1), len(multi))[1] == multi


def test_encoder_output_is_read_by_libbz2(harness, oracle_mod):
    for i, c in enumerate(sample_inputs(oracle_mod)):
        z = harness.compress(c)
        assert bz2.decompress(z) == c, i
        assert harness.decompress(z, len(c))[1] == c, i
        if len(c) < 800_000:
            assert len(z) == len(bz2.compress(c)), i          # same tables as libbz2 on single-block inputs
    multi = oracle_mod.gen_enwik_like(350_000, 5, 4096).tobytes() + oracle_mod.gen_uniform(150_000, 4).tobytes()
    assert bz2.decompress(harness.compress(multi, 1)) == multi


def test_decoder_rejects_damage(harness, oracle_mod):
    c = oracle_mod.gen_enwik_like(5000, 1, 512).tobytes()
    z = bytearray(bz2.compress(c))
    assert harness.decompress(bytes(z[:-5]), 6000)[0] == -2                  # truncated
    z[len(z) // 2] ^= 0x10
    assert harness.decompress(bytes(z), 6000)[0] == -1                       # CRC / structure
    assert harness.decompress(b"BZh9" + b"\x00" * 20, 100)[0] < 0
    assert harness.decompress(b"not bzip2", 100)[0] < 0


def packed(y):
    b = bytearray(struct.pack("<Q", abs(y)))
    if y < 0:
        b[7] |= 0x80
    return bytes(b)


def reference_style_patch(oracle_mod, old, new):
    """What Diff.Create writes, built from the oracle's raw streams and libbz2's framing."""
    sa = oracle_mod.divsufsort(old)
    ctrl, diff, extra, _ = oracle_mod.bsdiff_scan(old, sa, new)
    c = b"".join(packed(int(v)) for v in ctrl.reshape(-1))
    zc, zd, ze = bz2.compress(c), bz2.compress(diff.tobytes()), bz2.compress(extra.tobytes())
    return b"BSDIFF40" + packed(len(zc)) + packed(len(zd)) + packed(new.size) + zc + zd + ze


@pytest.mark.parametrize("size", [0, 1, 512, 999, 1024, 4096])
def test_patch_apply_roundtrip_like_BsDiffTests(backend_lib, oracle_mod, size):
    from deltaq_amd import Patch
    old = oracle_mod.net_random_bytes(size)
    for new in (old.copy(), oracle_mod.gen_uniform(size, 5), np.concatenate([old[size // 2:], old[:size // 3]])):
        patch = reference_style_patch(oracle_mod, old, new)
        assert Patch.Apply(old, patch) == new.tobytes()


def test_patch_apply_rejects_corrupt_patches(backend_lib, oracle_mod):
    from deltaq_amd import Patch
    old = oracle_mod.gen_enwik_like(20000, 1, 1024)
    new = np.concatenate([old[:5000], oracle_mod.gen_uniform(300, 2), old[7000:]])
    patch = reference_style_patch(oracle_mod, old, new)
    assert Patch.Apply(old, patch) == new.tobytes()
    for bad in (patch[:31], b"BSDIFF41" + patch[8:], patch[:8] + packed(-5) + patch[16:], patch[:60],
                patch[:24] + packed(new.size + 10) + patch[32:]):
        with pytest.raises(ValueError, match="Corrupt patch"):
            Patch.Apply(old, bad)
    with pytest.raises(ValueError, match="Corrupt patch"):
        Patch.Apply(old[:100], patch)                                       # the wrong old file: reads past its end
