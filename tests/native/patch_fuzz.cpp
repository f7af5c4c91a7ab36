// patch_fuzz.cpp -- TEST INFRASTRUCTURE: hostile patches against the product's Patch.Apply (deltaq_amd/csrc/dq_bspatch.h),
// built as a standalone program under -fsanitize=address,undefined (tests/test_patch_hardening.py).  A patch is
// untrusted input: whatever its bytes say, apply_patch must return a code -- no out-of-bounds access, no signed
// overflow, no allocation beyond what the new file's size justifies.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all tests/native/patch_fuzz.cpp -o patch_fuzz
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "../../deltaq_amd/csrc/dq_bspatch.h"

using namespace dq;

// suffix array by prefix doubling (the bomb below is one long run: a comparison sort of its suffixes would be quadratic)
static int doubling_sorter(const uint8_t *t, int64_t n, int32_t *sa)
{
    std::vector<int64_t> rank((size_t)n), tmp((size_t)n);
    std::iota(sa, sa + n, 0);
    for (int64_t i = 0; i < n; ++i) rank[(size_t)i] = t[i];
    for (int64_t h = 1;; h *= 2) {
        auto key2 = [&](int32_t a) { return a + h < n ? rank[(size_t)(a + h)] : (int64_t)-1; };
        auto less = [&](int32_t a, int32_t b) {
            return rank[(size_t)a] != rank[(size_t)b] ? rank[(size_t)a] < rank[(size_t)b] : key2(a) < key2(b);
        };
        std::sort(sa, sa + n, less);
        tmp[(size_t)sa[0]] = 0;
        for (int64_t i = 1; i < n; ++i) tmp[(size_t)sa[i]] = tmp[(size_t)sa[i - 1]] + (less(sa[i - 1], sa[i]) ? 1 : 0);
        rank = tmp;
        if (n == 0 || rank[(size_t)sa[n - 1]] == n - 1) break;
    }
    return 0;
}

static std::vector<uint8_t> zip(const std::vector<uint8_t> &v)
{
    std::vector<uint8_t> z;
    if (bz2::bz2_compress(v.data(), v.size(), z, doubling_sorter, 9) != 0) { fprintf(stderr, "compress failed\n"); exit(2); }
    return z;
}

static std::vector<uint8_t> make_patch(const std::vector<int64_t> &triples, const std::vector<uint8_t> &diff,
                                       const std::vector<uint8_t> &extra, int64_t new_size)
{
    std::vector<uint8_t> c(triples.size() * 8);
    for (size_t i = 0; i < triples.size(); ++i) bsdiff::write_packed_long(&c[i * 8], triples[i]);
    const std::vector<uint8_t> zc = zip(c), zd = zip(diff), ze = zip(extra);
    std::vector<uint8_t> p(32);
    bsdiff::write_packed_long(&p[0], bsdiff::kSignature);
    bsdiff::write_packed_long(&p[8], (int64_t)zc.size());
    bsdiff::write_packed_long(&p[16], (int64_t)zd.size());
    bsdiff::write_packed_long(&p[24], new_size);
    p.insert(p.end(), zc.begin(), zc.end());
    p.insert(p.end(), zd.begin(), zd.end());
    p.insert(p.end(), ze.begin(), ze.end());
    return p;
}

static int g_failures = 0;
#define EXPECT(cond, what)                                                  \
    do {                                                                    \
        if (!(cond)) { fprintf(stderr, "FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++g_failures; } \
    } while (0)

static int apply(const std::vector<uint8_t> &old, const std::vector<uint8_t> &patch, std::vector<uint8_t> &out)
{
    int64_t len = -1;
    int rc = bsdiff::apply_patch(old.data(), (int64_t)old.size(), patch.data(), (int64_t)patch.size(), nullptr, 0, &len);
    if (rc != 0) return rc;
    if (len > (1 << 26)) return -9;                       // (the harness does not allocate what a hostile header claims)
    out.assign((size_t)len, 0);
    return bsdiff::apply_patch(old.data(), (int64_t)old.size(), patch.data(), (int64_t)patch.size(), out.data(), len, &len);
}

int main()
{
    std::vector<uint8_t> old(5000), out;
    uint64_t x = 12345;
    auto rnd = [&]() { x += 0x9E3779B97F4A7C15ull; uint64_t z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                       z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    for (auto &b : old) b = (uint8_t)(rnd() % 7);
    constexpr int64_t kMax = INT64_MAX;

    // a well-formed patch first: 100 bytes added from old[10..], 5 extra, back by 50, 20 more added
    {
        std::vector<uint8_t> diff(120, 1), extra{9, 8, 7, 6, 5};
        const std::vector<uint8_t> p = make_patch({100, 5, -50, 20, 0, 0}, diff, extra, 125);
        EXPECT(apply(old, p, out) == 0 && out.size() == 125, "valid patch applies");
        EXPECT(out[0] == (uint8_t)(old[0] + 1) && out[100] == 9 && out[105] == (uint8_t)(old[50] + 1), "valid patch content");
    }
    // 1. header lengths whose sum wraps (ctrl_len = diff_len = 2^62 was accepted)
    {
        std::vector<uint8_t> p(64, 0);
        bsdiff::write_packed_long(&p[0], bsdiff::kSignature);
        bsdiff::write_packed_long(&p[8], (int64_t)1 << 62);
        bsdiff::write_packed_long(&p[16], (int64_t)1 << 62);
        bsdiff::write_packed_long(&p[24], 10);
        EXPECT(apply(old, p, out) == -1, "wrapping header lengths are corrupt");
        bsdiff::write_packed_long(&p[8], kMax);
        bsdiff::write_packed_long(&p[16], 1);
        EXPECT(apply(old, p, out) == -1, "ctrl_len = INT64_MAX is corrupt");
        bsdiff::write_packed_long(&p[8], 16);
        bsdiff::write_packed_long(&p[16], kMax);
        EXPECT(apply(old, p, out) == -1, "diff_len = INT64_MAX is corrupt");
    }
    // 2. triples that wrap the old-file position / the bounds sums
    {
        std::vector<uint8_t> diff(16, 0), extra(16, 0);
        EXPECT(apply(old, make_patch({0, 0, kMax, 1, 0, 0}, diff, extra, 1), out) == -1, "seek to INT64_MAX then read");
        EXPECT(apply(old, make_patch({0, 0, kMax, 0, 0, kMax, 1, 0, 0}, diff, extra, 1), out) == -1, "two seeks that wrap");
        EXPECT(apply(old, make_patch({kMax, 0, 0}, diff, extra, 4), out) == -1, "add = INT64_MAX");
        EXPECT(apply(old, make_patch({1, kMax, 0}, diff, extra, 4), out) == -1, "copy = INT64_MAX");
        EXPECT(apply(old, make_patch({2, 0, -kMax, 2, 0, 0}, diff, extra, 4), out) == -1, "seek far below zero");
        EXPECT(apply(old, make_patch({2, 0, -3, 2, 0, 0}, diff, extra, 4), out) == -1, "seek just below zero");
        EXPECT(apply(old, make_patch({-1, 0, 0, 4, 0, 0}, diff, extra, 4), out) == -1, "negative add");
        EXPECT(apply(old, make_patch({0, 0, 5001, 0, 4, 0}, diff, extra, 4), out) == 0, "past the end without reading is fine");
        EXPECT(apply(old, make_patch({0, 0, 4999, 2, 0, 0}, diff, extra, 2), out) == -1, "read across the end of old");
    }
    // 3. decompression bombs: streams that decode to far more than the new file could use are cut, not expanded
    {
        std::vector<uint8_t> big(4u << 20, 0);                    // 4 MiB of zeros -> a few dozen bytes of bzip2
        const std::vector<uint8_t> z = zip(big);
        EXPECT(z.size() < 4096, "bomb is small");
        std::vector<uint8_t> dec;
        const int rc = bz2::bz2_decompress(z.data(), z.size(), dec, 1000);
        EXPECT(rc == bz2::kTooLong && dec.size() == 1000, "decoder stops at its limit");
        std::vector<uint8_t> diff(8, 0);
        const std::vector<uint8_t> p = make_patch({8, 0, 0}, big, diff, 8);           // diff stream = the bomb
        EXPECT(apply(old, p, out) == 0 && out.size() == 8 && out[3] == old[3], "bomb in the diff stream: first newSize bytes used");
        std::vector<uint8_t> p2 = make_patch({4, 4, 0}, diff, big, 8);                 // extra stream = the bomb
        EXPECT(apply(old, p2, out) == 0, "bomb in the extra stream");
    }
    // 4. mutations of a valid patch: any return code will do, the sanitizers judge the rest
    {
        std::vector<uint8_t> nw(6000);
        for (size_t i = 0; i < nw.size(); ++i) nw[i] = i < 5000 ? (uint8_t)(old[i] + (rnd() % 50 == 0)) : (uint8_t)rnd();
        std::vector<uint8_t> diff(5000), extra(nw.begin() + 5000, nw.end());
        for (size_t i = 0; i < 5000; ++i) diff[i] = (uint8_t)(nw[i] - old[i]);
        const std::vector<uint8_t> good = make_patch({5000, 1000, 0}, diff, extra, 6000);
        EXPECT(apply(old, good, out) == 0 && out == nw, "fuzz seed applies");
        int ok = 0;
        for (int it = 0; it < 4000; ++it) {
            std::vector<uint8_t> p = good;
            const int edits = 1 + (int)(rnd() % 4);
            for (int e = 0; e < edits; ++e) {
                const uint64_t r = rnd();
                const size_t at = (r >> 8) % p.size();
                switch (r & 3) {
                    case 0: p[at] ^= (uint8_t)(1u << ((r >> 40) & 7)); break;
                    case 1: p[at] = (uint8_t)(r >> 48); break;
                    case 2: p.resize(at + 1); break;
                    default: if (at < 32) p[at] = (uint8_t)(r >> 48); else p.insert(p.begin() + (long)at, (uint8_t)(r >> 48)); break;
                }
            }
            ok += apply(old, p, out) == 0;
        }
        printf("mutated patches accepted: %d of 4000\n", ok);
    }
    if (g_failures) { fprintf(stderr, "%d failures\n", g_failures); return 1; }
    printf("ok\n");
    return 0;
}
