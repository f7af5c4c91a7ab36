// bz2_harness.cpp -- TEST INFRASTRUCTURE: exposes the product's host-side bzip2 codec (deltaq_amd/csrc/dq_bz2.h) to the
// CPU tests through ctypes.  The encoder's Burrows-Wheeler transform needs a suffix array of block+block; the
// product takes it from the MI355X sorter, this harness from a naive comparison sort (small inputs only), so that
// the format logic of the encoder can be checked against libbz2 on a machine without a GPU.
//   g++ -O2 -std=c++17 -fPIC -shared tests/native/bz2_harness.cpp -o tests/native/libbz2_harness.so
#include <cstdlib>
#include <numeric>

#include "../../deltaq_amd/csrc/dq_bz2.h"
#include "../../deltaq_amd/csrc/dq_bspatch.h"

static int naive_sorter(const uint8_t *t, int64_t n, int32_t *sa)
{
    std::iota(sa, sa + n, 0);
    std::sort(sa, sa + n, [&](int32_t a, int32_t b) {
        const int64_t la = n - a, lb = n - b, m = la < lb ? la : lb;
        const int c = memcmp(t + a, t + b, (size_t)m);
        return c != 0 ? c < 0 : la < lb;
    });
    return 0;
}

extern "C" {

int64_t t_bz2_prepass_span(const uint8_t *src, int64_t n, int32_t level, int64_t max_step, uint32_t seed, uint8_t *rle, int64_t cap,
                           uint32_t *lens, uint32_t *crcs, int64_t max_blocks, int64_t span);

// returns the compressed length, or a negative code; cap = capacity of out
int64_t t_bz2_compress(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap, int32_t level)
{
    std::vector<uint8_t> v;
    const int rc = dq::bz2::bz2_compress(src, (size_t)n, v, naive_sorter, level);
    if (rc != 0) return rc;
    if ((int64_t)v.size() > cap) return -100;
    memcpy(out, v.data(), v.size());
    return (int64_t)v.size();
}

// the same stream fed piece by piece as a producer would hand it over (pieces of 0 .. max_step bytes, seeded): the
// pre-pass follows the pieces, full blocks are encoded on threads of their own while it carries on
int64_t t_bz2_compress_fed(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap, int32_t level, int64_t max_step, uint32_t seed)
{
    std::vector<uint8_t> v;
    dq::bz2::StreamEncoder enc(naive_sorter, level);
    uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
    int64_t upto = 0;
    while (upto < n) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        upto = std::min<int64_t>(n, upto + (int64_t)((x >> 33) % (uint64_t)(max_step + 1)));
        enc.feed(src, (size_t)upto, false);
    }
    enc.feed(src, (size_t)n, true);
    const int rc = enc.finish(v);
    if (rc != 0) return rc;
    if ((int64_t)v.size() > cap) return -100;
    memcpy(out, v.data(), v.size());
    return (int64_t)v.size();
}

// the pre-pass alone, fed in pieces: run-length coded blocks back to back in `rle` (capacity cap), their lengths and
// CRCs in lens / crcs (capacity max_blocks); returns the number of blocks, or a negative code
int64_t t_bz2_prepass(const uint8_t *src, int64_t n, int32_t level, int64_t max_step, uint32_t seed, uint8_t *rle, int64_t cap,
                      uint32_t *lens, uint32_t *crcs, int64_t max_blocks)
{
    return t_bz2_prepass_span(src, n, level, max_step, seed, rle, cap, lens, crcs, max_blocks, 0);
}

// ... with a bound on the bytes of the stream one block covers (0: none, the plain definition; the product's default is
// dq::bz2::kBlockSpan)
int64_t t_bz2_block_span(void) { return (int64_t)dq::bz2::kBlockSpan; }
int64_t t_bz2_prepass_span(const uint8_t *src, int64_t n, int32_t level, int64_t max_step, uint32_t seed, uint8_t *rle, int64_t cap,
                           uint32_t *lens, uint32_t *crcs, int64_t max_blocks, int64_t span)
{
    dq::bz2::StreamEncoder enc(naive_sorter, level, span > 0 ? (size_t)span : 0);
    enc.hold_blocks();
    uint64_t x = seed * 0x9e3779b97f4a7c15ull + 1;
    int64_t upto = 0;
    while (upto < n) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        upto = std::min<int64_t>(n, upto + (int64_t)((x >> 33) % (uint64_t)(max_step + 1)));
        enc.feed(src, (size_t)upto, false);
    }
    enc.feed(src, (size_t)n, true);
    std::vector<uint8_t> r;
    std::vector<uint32_t> l, c;
    enc.copy_blocks(r, l, c);
    if ((int64_t)r.size() > cap || (int64_t)l.size() > max_blocks) return -100;
    if (!r.empty()) memcpy(rle, r.data(), r.size());
    for (size_t k = 0; k < l.size(); ++k) { lens[k] = l[k]; crcs[k] = c[k]; }
    return (int64_t)l.size();
}

// limit < 0: no bound on the decoded size
int64_t t_bz2_decompress_limit(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap, int64_t limit, int32_t *code)
{
    std::vector<uint8_t> v;
    *code = dq::bz2::bz2_decompress(src, (size_t)n, v, limit < 0 ? (size_t)-1 : (size_t)limit);
    if ((int64_t)v.size() > cap) return -100;
    if (!v.empty()) memcpy(out, v.data(), v.size());
    return (int64_t)v.size();
}

// Patch.Apply as the product runs it (dq_bspatch.h): 0, -1 "Corrupt patch", -2 output buffer too small
int32_t t_bspatch_apply(const uint8_t *old_data, int64_t n, const uint8_t *patch, int64_t plen, uint8_t *out, int64_t cap,
                        int64_t *out_len)
{
    return dq::bsdiff::apply_patch(old_data, n, patch, plen, out, cap, out_len);
}

// the block CRC folded 8 bytes a step / on several threads, against the bit-by-bit definition
uint32_t t_crc_bitwise(const uint8_t *p, int64_t n)
{
    uint32_t crc = 0xffffffffu;
    for (int64_t i = 0; i < n; ++i) {
        crc ^= (uint32_t)p[i] << 24;
        for (int k = 0; k < 8; ++k) crc = (crc & 0x80000000u) ? (crc << 1) ^ 0x04c11db7u : crc << 1;
    }
    return ~crc;
}
uint32_t t_crc_sliced(const uint8_t *p, int64_t n) { return ~dq::bz2::crc_update(0xffffffffu, p, (size_t)n); }
uint32_t t_crc_mt(const uint8_t *p, int64_t n) { return ~dq::bz2::crc_update_mt(0xffffffffu, p, (size_t)n); }

int64_t t_bz2_decompress(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap)
{
    std::vector<uint8_t> v;
    const int rc = dq::bz2::bz2_decompress(src, (size_t)n, v);
    if (rc != 0) return rc;
    if ((int64_t)v.size() > cap) return -100;
    if (!v.empty()) memcpy(out, v.data(), v.size());
    return (int64_t)v.size();
}

}
