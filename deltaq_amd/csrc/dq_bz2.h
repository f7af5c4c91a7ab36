// dq_bz2.h -- bzip2 streams for the BSDIFF40 container (host code).
//
// The reference frames the three parts of a patch with SharpZipLib's BZip2OutputStream / BZip2InputStream
// (src/DeltaQ.BsDiff/Diff.cs:15-18, DeltaQ.BsDiff.csproj:42; third-party, not vendored).  What has to match is the
// FORMAT (bzip2 1.0: any conforming decoder must read what is written here, and what any conforming encoder
// wrote must be read here), not SharpZipLib's byte stream -- the reference itself pins patches only by round
// trips (BsDiffTests.cs:30-78).  tests/test_bz2_container.py checks both directions against libbz2 (Python's bz2).
//
//   bz2_decompress   the whole decoder (tables, MTF / RUNA-RUNB, inverse BWT, run-length, CRCs): pure host code
//   bz2_compress     run-length pre-pass, MTF / RUNA-RUNB, up to 6 Huffman tables refined over 50-symbol groups
//                    (the usual 4 iterations), canonical codes; the Burrows-Wheeler transform of a block is taken
//                    from a suffix array of block+block supplied by the caller -- in this library the MI355X
//                    suffix sorter itself (rotations i < j compare like the suffixes of block+block at i, j over
//                    their first n characters; rotations that are equal as strings carry equal last characters,
//                    so their mutual order does not matter)
#pragma once
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <thread>
#include <emmintrin.h>
#include <vector>

namespace dq {
namespace bz2 {

constexpr int kMaxAlpha = 258;
constexpr int kMaxCodeLen = 20;           // what a decoder must accept
constexpr int kEncCodeLen = 17;           // what this encoder produces (as libbz2)
constexpr int kGroupSize = 50;
constexpr int kMaxGroups = 6;
constexpr int kMaxSelectors = 2 + 900000 / kGroupSize;
constexpr uint64_t kBlockMagic = 0x314159265359ull, kEndMagic = 0x177245385090ull;

// bzip2's CRC-32: polynomial 0x04c11db7, most significant bit first.  v[0] is the classic byte table; v[k][i] is the
// CRC of byte i followed by k zero bytes, which lets crc_update() fold 8 input bytes per step ("slicing by 8") --
// the diff stream of two similar 16 MiB files is 16 MiB of mostly zeros, and a byte-at-a-time CRC over it cost
// more than the device's suffix sort of the file.
struct CrcTable {
    uint32_t v[8][256];
    constexpr CrcTable() : v{}
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i << 24;
            for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? (c << 1) ^ 0x04c11db7u : c << 1;
            v[0][i] = c;
        }
        for (int k = 1; k < 8; ++k)
            for (uint32_t i = 0; i < 256; ++i) v[k][i] = (v[k - 1][i] << 8) ^ v[0][v[k - 1][i] >> 24];
    }
};
// built at compile time: nothing to race on when several threads frame their first patch at once
inline constexpr CrcTable kCrcTable{};
inline const uint32_t *crc_table() { return kCrcTable.v[0]; }

// ---- the register update is linear over GF(2): the state after A || B from state s is
// Z^{len(B)}(state after A from s)  xor  (state after B from 0), Z = "one zero byte" as a 32 x 32 bit matrix (the
// construction of zlib's crc32_combine, for this polynomial and bit order).
inline uint32_t gf2_times(const uint32_t *mat, uint32_t vec)
{
    uint32_t sum = 0;
    for (int i = 0; vec; vec >>= 1, ++i)
        if (vec & 1u) sum ^= mat[i];
    return sum;
}

// Z^(2^k) for k = 0 .. 47, built once
struct ZeroBytePowers {
    uint32_t m[48][32];
    ZeroBytePowers()
    {
        for (int i = 0; i < 32; ++i) {                       // column i: what one zero byte makes of the state 1 << i
            const uint32_t c = 1u << i;
            m[0][i] = (c << 8) ^ kCrcTable.v[0][c >> 24];
        }
        for (int k = 1; k < 48; ++k)
            for (int i = 0; i < 32; ++i) m[k][i] = gf2_times(m[k - 1], m[k - 1][i]);
    }
};

// state after nbytes zero bytes, from `state`
inline uint32_t crc_shift(uint32_t state, uint64_t nbytes)
{
    static const ZeroBytePowers zp;
    for (int k = 0; nbytes && k < 48; nbytes >>= 1, ++k)
        if (nbytes & 1u) state = gf2_times(zp.m[k], state);
    return state;
}

// (a stretch of >= 64 zero bytes -- the diff stream of two similar files is little else -- is stepped over with
// crc_shift: a few dozen table-free operations instead of one table lookup per byte)
inline uint32_t crc_update(uint32_t crc, const uint8_t *p, size_t n)
{
    const auto &T = kCrcTable.v;
    size_t i = 0;
    while (i + 8 <= n) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        if (w == 0) {
            size_t j = i + 8;
            while (j + 8 <= n) {
                memcpy(&w, p + j, 8);
                if (w) break;
                j += 8;
            }
            if (j - i >= 64) {
                crc = crc_shift(crc, j - i);
                i = j;
                continue;
            }
        }
        const uint32_t a = crc ^ (((uint32_t)p[i] << 24) | ((uint32_t)p[i + 1] << 16) | ((uint32_t)p[i + 2] << 8) | p[i + 3]);
        crc = T[7][a >> 24] ^ T[6][(a >> 16) & 255] ^ T[5][(a >> 8) & 255] ^ T[4][a & 255] ^
              T[3][p[i + 4]] ^ T[2][p[i + 5]] ^ T[1][p[i + 6]] ^ T[0][p[i + 7]];
        i += 8;
    }
    for (; i < n; ++i) crc = (crc << 8) ^ T[0][(crc >> 24) ^ p[i]];
    return crc;
}

// ---- the same CRC over a long buffer on several threads
inline uint32_t crc_update_mt(uint32_t crc, const uint8_t *p, size_t n)
{
    constexpr size_t kMinPart = (size_t)1 << 20;
    const unsigned parts = (unsigned)std::min<size_t>(4, n / kMinPart);
    if (parts < 2) return crc_update(crc, p, n);
    const size_t per = n / parts;
    uint32_t raw[4] = {0, 0, 0, 0};
    {
        std::vector<std::thread> ts;
        for (unsigned k = 1; k < parts; ++k) {
            const uint8_t *q = p + k * per;
            const size_t len = k + 1 == parts ? n - k * per : per;
            try { ts.emplace_back([&raw, k, q, len] { raw[k] = crc_update(0, q, len); }); }
            catch (...) { raw[k] = crc_update(0, q, len); }
        }
        raw[0] = crc_update(crc, p, per);
        for (std::thread &t : ts) t.join();
    }
    uint32_t s = raw[0];
    for (unsigned k = 1; k < parts; ++k) {
        const size_t len = k + 1 == parts ? n - k * per : per;
        s = crc_shift(s, len) ^ raw[k];
    }
    return s;
}

// ------------------------------------------------------------------ decoder
struct BitReader {
    const uint8_t *p;
    size_t n, pos = 0;
    uint64_t buf = 0;
    int have = 0;
    bool overrun = false;
    BitReader(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    uint32_t bits(int k)                 // k <= 32
    {
        while (have < k) {
            if (pos >= n) { overrun = true; buf <<= 8; have += 8; continue; }
            buf = (buf << 8) | p[pos++];
            have += 8;
        }
        const uint32_t v = (uint32_t)((buf >> (have - k)) & ((k == 32) ? 0xffffffffull : ((1ull << k) - 1)));
        have -= k;
        return v;
    }
    uint32_t bit() { return bits(1); }
};

enum { kOk = 0, kCorrupt = -1, kTruncated = -2, kTooLong = -3 };

// Decodes one whole stream (possibly several concatenated streams, as bzip2 allows).  max_out bounds the decoded
// size: a block expands up to ~50x per run-length level and streams concatenate, so a patch of a few KB could
// otherwise ask for gigabytes; the caller knows how much it can use (Patch.Apply stops reading at newSize).
inline int bz2_decompress(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t max_out = (size_t)-1)
{
    BitReader br(src, n);
    std::vector<uint32_t> tt;
    bool first_stream = true;
    for (;;) {
        if (br.pos >= n && br.have < 8) { return first_stream ? kTruncated : kOk; }
        if (br.bits(8) != 'B' || br.bits(8) != 'Z' || br.bits(8) != 'h') return first_stream ? kCorrupt : kOk;
        const int level = (int)br.bits(8) - '0';
        if (level < 1 || level > 9) return kCorrupt;
        const uint32_t block_max = (uint32_t)level * 100000u;
        uint32_t combined = 0;
        for (;;) {
            const uint64_t magic = ((uint64_t)br.bits(24) << 24) | br.bits(24);
            if (br.overrun) return kTruncated;
            if (magic == kEndMagic) {
                const uint32_t stored = br.bits(32);
                if (br.overrun) return kTruncated;
                if (stored != combined) return kCorrupt;
                br.have -= br.have % 8;                       // streams are byte aligned
                break;
            }
            if (magic != kBlockMagic) return kCorrupt;
            const uint32_t block_crc = br.bits(32);
            if (br.bit()) return kCorrupt;                    // randomised blocks: deprecated, never written
            const uint32_t orig_ptr = br.bits(24);
            // symbols in use
            uint8_t seq_to_unseq[256];
            int n_in_use = 0;
            {
                const uint32_t used16 = br.bits(16);
                for (int i = 0; i < 16; ++i) {
                    if (used16 & (0x8000u >> i)) {
                        const uint32_t m16 = br.bits(16);
                        for (int j = 0; j < 16; ++j)
                            if (m16 & (0x8000u >> j)) seq_to_unseq[n_in_use++] = (uint8_t)(i * 16 + j);
                    }
                }
            }
            if (n_in_use == 0) return kCorrupt;
            const int alpha = n_in_use + 2;
            const int n_groups = (int)br.bits(3);
            if (n_groups < 2 || n_groups > kMaxGroups) return kCorrupt;
            const int n_sel = (int)br.bits(15);
            if (n_sel < 1) return kCorrupt;
            std::vector<uint8_t> selector((size_t)n_sel);
            {
                uint8_t pos[kMaxGroups];
                for (int i = 0; i < n_groups; ++i) pos[i] = (uint8_t)i;
                for (int i = 0; i < n_sel; ++i) {
                    int j = 0;
                    while (br.bit()) { if (++j >= n_groups) return kCorrupt; }
                    const uint8_t v = pos[j];
                    for (; j > 0; --j) pos[j] = pos[j - 1];
                    pos[0] = v;
                    selector[(size_t)i] = v;
                }
            }
            // coding tables
            uint8_t len[kMaxGroups][kMaxAlpha];
            for (int t = 0; t < n_groups; ++t) {
                int curr = (int)br.bits(5);
                for (int i = 0; i < alpha; ++i) {
                    for (;;) {
                        if (curr < 1 || curr > kMaxCodeLen) return kCorrupt;
                        if (!br.bit()) break;
                        curr += br.bit() ? -1 : 1;
                    }
                    len[t][i] = (uint8_t)curr;
                }
            }
            if (br.overrun) return kTruncated;
            int32_t limit[kMaxGroups][kMaxCodeLen + 2], base[kMaxGroups][kMaxCodeLen + 2], perm[kMaxGroups][kMaxAlpha];
            int min_len[kMaxGroups];
            for (int t = 0; t < n_groups; ++t) {
                int mn = 32, mx = 0;
                for (int i = 0; i < alpha; ++i) { mn = std::min<int>(mn, len[t][i]); mx = std::max<int>(mx, len[t][i]); }
                min_len[t] = mn;
                int pp = 0;
                for (int l = mn; l <= mx; ++l)
                    for (int i = 0; i < alpha; ++i)
                        if (len[t][i] == l) perm[t][pp++] = i;
                int cnt[kMaxCodeLen + 2] = {0};
                for (int i = 0; i < alpha; ++i) cnt[len[t][i]]++;
                int32_t code = 0, idx = 0;
                for (int l = 0; l <= kMaxCodeLen + 1; ++l) { limit[t][l] = -1; base[t][l] = 0; }
                for (int l = mn; l <= mx; ++l) {
                    base[t][l] = idx - code;                    // perm index = code + base
                    code += cnt[l];
                    idx += cnt[l];
                    limit[t][l] = code - 1;                     // largest code of this length
                    code <<= 1;
                }
                for (int l = mx + 1; l <= kMaxCodeLen + 1; ++l) limit[t][l] = 0x7fffffff;      // forces termination
            }
            // MTF / run-length decoding into tt (the last column of the sorted rotations)
            tt.clear();
            tt.reserve(block_max);
            uint32_t unzftab[256] = {0};
            uint8_t mtf[256];
            for (int i = 0; i < 256; ++i) mtf[i] = (uint8_t)i;
            const int eob = n_in_use + 1;
            int group_no = -1, group_pos = 0, t = 0;
            int64_t run = 0, run_weight = 1;                   // pending RUNA/RUNB run: its length so far
            bool in_run = false;
            auto flush_run = [&]() -> bool {
                if (!in_run) return true;
                if (tt.size() + (size_t)run > block_max) return false;
                const uint8_t ch = seq_to_unseq[mtf[0]];
                unzftab[ch] += (uint32_t)run;
                tt.insert(tt.end(), (size_t)run, (uint32_t)ch);
                run = 0;
                run_weight = 1;
                in_run = false;
                return true;
            };
            for (;;) {
                if (group_pos == 0) {
                    if (++group_no >= n_sel) return kCorrupt;
                    group_pos = kGroupSize;
                    t = selector[(size_t)group_no];
                }
                --group_pos;
                int l = min_len[t];
                int32_t code = (int32_t)br.bits(l);
                while (l <= kMaxCodeLen && code > limit[t][l]) { code = (code << 1) | (int32_t)br.bit(); ++l; }
                if (l > kMaxCodeLen || br.overrun) return br.overrun ? kTruncated : kCorrupt;
                const int32_t pi = code + base[t][l];
                if (pi < 0 || pi >= alpha) return kCorrupt;
                const int sym = perm[t][pi];
                if (sym == eob) break;
                if (sym <= 1) {                                // RUNA = 0, RUNB = 1: digits 1 / 2 of a bijective base-2 number
                    if (run_weight > (1 << 21)) return kCorrupt;
                    in_run = true;
                    run += (int64_t)(sym + 1) * run_weight;
                    run_weight <<= 1;
                    continue;
                }
                if (!flush_run()) return kCorrupt;
                if (tt.size() >= block_max) return kCorrupt;
                const int idx = sym - 1;                       // MTF position
                if (idx >= n_in_use) return kCorrupt;
                const uint8_t v = mtf[idx];
                memmove(mtf + 1, mtf, (size_t)idx);
                mtf[0] = v;
                const uint8_t ch = seq_to_unseq[v];
                unzftab[ch]++;
                tt.push_back((uint32_t)ch);
            }
            if (!flush_run()) return kCorrupt;
            const uint32_t nblock = (uint32_t)tt.size();
            if (orig_ptr >= nblock) return kCorrupt;
            // inverse BWT: tt[cftab[ch]++] |= i << 8, then follow the chain from orig_ptr
            uint32_t cftab[257];
            cftab[0] = 0;
            for (int i = 0; i < 256; ++i) cftab[i + 1] = cftab[i] + unzftab[i];
            for (uint32_t i = 0; i < nblock; ++i) {
                const uint8_t ch = (uint8_t)(tt[i] & 0xff);
                tt[cftab[ch]++] |= i << 8;
            }
            // un-run-length (4 equal bytes + a count byte) while following the chain; CRC of the block's output after it
            const size_t out0 = out.size();
            uint32_t tpos = tt[orig_ptr] >> 8;
            int same = 0, prev = -1;
            for (uint32_t i = 0; i < nblock; ++i) {
                const uint32_t e = tt[tpos];
                const uint8_t ch = (uint8_t)(e & 0xff);
                tpos = e >> 8;
                if (same == 4) {                               // ch is a repeat count
                    if (out.size() + ch > max_out) { out.insert(out.end(), max_out - out.size(), (uint8_t)prev); return kTooLong; }
                    out.insert(out.end(), (size_t)ch, (uint8_t)prev);
                    same = 0;
                    prev = -1;
                    continue;
                }
                if (out.size() >= max_out) return kTooLong;
                out.push_back(ch);
                same = (ch == prev) ? same + 1 : 1;
                prev = ch;
            }
            uint32_t crc = crc_update_mt(0xffffffffu, out.data() + out0, out.size() - out0);
            crc = ~crc;
            if (crc != block_crc) return kCorrupt;
            combined = ((combined << 1) | (combined >> 31)) ^ crc;
        }
        first_stream = false;
    }
}

// ------------------------------------------------------------------ encoder
struct BitWriter {
    std::vector<uint8_t> &out;
    uint64_t buf = 0;
    int have = 0;
    uint64_t total = 0;                  // bits written so far
    explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
    void bits(int k, uint32_t v)         // k <= 32
    {
        buf = (buf << k) | (k == 32 ? (uint64_t)v : ((uint64_t)v & ((1ull << k) - 1)));
        have += k;
        total += (uint64_t)k;
        while (have >= 8) { out.push_back((uint8_t)(buf >> (have - 8))); have -= 8; }
    }
    void flush() { if (have > 0) { out.push_back((uint8_t)(buf << (8 - have))); have = 0; } }
    // nbits bits of src (the most significant bit of src[0] first) behind what has been written: whole bytes 8 at a time
    // behind the pending bits (the blocks of a stream are coded on their own and strung together here -- a byte a call,
    // 4 MB of them took 6 ms)
    void append(const uint8_t *src, uint64_t nbits)
    {
        const uint64_t whole = nbits / 8;
        if (whole) {
            const size_t at = out.size();
            out.resize(at + whole);
            uint8_t *o = out.data() + at;
            if (have == 0) {
                memcpy(o, src, whole);
            } else {
                const int k = have;                       // 1 .. 7 pending bits in front of every byte
                uint64_t carry = buf & ((1ull << k) - 1);
                uint64_t q = 0;
                for (; q + 8 <= whole; q += 8) {
                    uint64_t v;
                    memcpy(&v, src + q, 8);
                    v = __builtin_bswap64(v);
                    const uint64_t w = __builtin_bswap64((carry << (64 - k)) | (v >> k));
                    memcpy(o + q, &w, 8);
                    carry = v & ((1ull << k) - 1);
                }
                for (; q < whole; ++q) {
                    const uint64_t v = (carry << 8) | src[q];
                    o[q] = (uint8_t)(v >> k);
                    carry = v & ((1ull << k) - 1);
                }
                buf = carry;
            }
            total += whole * 8;
        }
        const int rest = (int)(nbits - whole * 8);
        if (rest > 0) bits(rest, (uint32_t)src[whole] >> (8 - rest));
    }
};

// Code lengths (<= max_len) of a Huffman code for freq[0..alpha): package of the classic libbz2 approach --
// build the tree, and while it is too deep flatten the frequencies and rebuild.
inline void make_code_lengths(uint8_t *len, const int32_t *freq, int alpha, int max_len)
{
    std::vector<int64_t> weight((size_t)alpha * 2 + 2);
    std::vector<int32_t> parent((size_t)alpha * 2 + 2), heap((size_t)alpha + 2);
    std::vector<int32_t> f(freq, freq + alpha);
    for (;;) {
        // weights carry the depth in their low 8 bits, as in libbz2
        for (int i = 0; i < alpha; ++i) weight[(size_t)i + 1] = (int64_t)(f[(size_t)i] == 0 ? 1 : f[(size_t)i]) << 8;
        int n_nodes = alpha, n_heap = 0;
        heap[0] = 0; weight[0] = 0; parent[0] = -2;
        auto up = [&](int z) {
            const int32_t tmp = heap[(size_t)z];
            while (weight[(size_t)tmp] < weight[(size_t)heap[(size_t)(z >> 1)]]) { heap[(size_t)z] = heap[(size_t)(z >> 1)]; z >>= 1; }
            heap[(size_t)z] = tmp;
        };
        auto down = [&](int z) {
            const int32_t tmp = heap[(size_t)z];
            for (;;) {
                int y = z << 1;
                if (y > n_heap) break;
                if (y < n_heap && weight[(size_t)heap[(size_t)y + 1]] < weight[(size_t)heap[(size_t)y]]) ++y;
                if (weight[(size_t)tmp] < weight[(size_t)heap[(size_t)y]]) break;
                heap[(size_t)z] = heap[(size_t)y];
                z = y;
            }
            heap[(size_t)z] = tmp;
        };
        for (int i = 1; i <= alpha; ++i) { parent[(size_t)i] = -1; heap[(size_t)++n_heap] = i; up(n_heap); }
        while (n_heap > 1) {
            const int32_t n1 = heap[1]; heap[1] = heap[(size_t)n_heap--]; down(1);
            const int32_t n2 = heap[1]; heap[1] = heap[(size_t)n_heap--]; down(1);
            ++n_nodes;
            parent[(size_t)n1] = parent[(size_t)n2] = n_nodes;
            const int64_t w1 = weight[(size_t)n1], w2 = weight[(size_t)n2];
            weight[(size_t)n_nodes] = ((w1 & ~0xffll) + (w2 & ~0xffll)) | (1 + std::max(w1 & 0xff, w2 & 0xff));
            parent[(size_t)n_nodes] = -1;
            heap[(size_t)++n_heap] = n_nodes;
            up(n_heap);
        }
        bool too_long = false;
        for (int i = 1; i <= alpha; ++i) {
            int j = 0, k = i;
            while (parent[(size_t)k] >= 0) { k = parent[(size_t)k]; ++j; }
            len[i - 1] = (uint8_t)j;
            if (j > max_len) too_long = true;
        }
        if (!too_long) return;
        for (int i = 0; i < alpha; ++i) f[(size_t)i] = 1 + (f[(size_t)i] == 0 ? 0 : f[(size_t)i]) / 2;
    }
}

// sa_of_doubled(text, n2, sa): suffix array (int32, n2 entries) of the n2 = 2 * nblock bytes block+block.
using DoubledSorter = std::function<int(const uint8_t *text, int64_t n2, int32_t *sa)>;

// One block: `blk` is the run-length coded data (nblock bytes), crc the CRC of the original bytes it stands for.
inline int compress_block(BitWriter &bw, const std::vector<uint8_t> &blk, uint32_t crc, const DoubledSorter &sorter)
{
    const int32_t nblock = (int32_t)blk.size();
    const bool trace = getenv("DQ_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    double t_ph[5] = {0, 0, 0, 0, 0};                       // sort, last column, MTF, tables, bits
    auto lap = [&](int k) { const auto t = now(); t_ph[k] += std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; };
    // ---- Burrows-Wheeler transform through the suffix array of block+block ----
    std::vector<uint8_t> doubled((size_t)nblock * 2);
    memcpy(doubled.data(), blk.data(), (size_t)nblock);
    memcpy(doubled.data() + nblock, blk.data(), (size_t)nblock);
    std::vector<int32_t> sa((size_t)nblock * 2);
    const int rc = sorter(doubled.data(), (int64_t)nblock * 2, sa.data());
    if (rc != 0) return rc;
    lap(0);
    std::vector<uint8_t> last((size_t)nblock);
    int32_t orig_ptr = -1, row = 0;
    for (int32_t i = 0; i < 2 * nblock; ++i) {
        const int32_t s = sa[(size_t)i];
        if (s >= nblock) continue;
        if (s == 0) orig_ptr = row;
        last[(size_t)row++] = blk[(size_t)(s == 0 ? nblock - 1 : s - 1)];
    }
    if (row != nblock || orig_ptr < 0) return -3;
    lap(1);
    // ---- symbols in use, MTF, RUNA / RUNB ----
    bool in_use[256] = {false};
    for (int32_t i = 0; i < nblock; ++i) in_use[blk[(size_t)i]] = true;
    uint8_t unseq_to_seq[256];
    int n_in_use = 0;
    for (int i = 0; i < 256; ++i) if (in_use[i]) unseq_to_seq[i] = (uint8_t)n_in_use++;
    const int eob = n_in_use + 1, alpha = n_in_use + 2;
    std::vector<uint16_t> mtfv;
    mtfv.reserve((size_t)nblock + 1);
    int32_t mtf_freq[kMaxAlpha] = {0};
    {
        // Move-to-front without the list: a symbol's place in it is the number of symbols used more recently, so every
        // symbol keeps the time of its last use (16 bits, renumbered before they run out) and a place is one sweep of
        // 16-bit compares over the alphabet, 8 a step -- ~30 instructions whatever the depth, no byte that was just
        // stored is loaded again.  (The classic list, searched and moved 8 or 16 entries a step, took 18 ns per symbol of
        // a block of random bytes -- average depth 128, every search reading what the last move wrote one byte off;
        // walking it byte by byte, as the classic coder does, 33 ms for a 900 kB block.)
        alignas(16) int16_t used[256 + 8] = {0};               // 0: not in the alphabet (never "more recent" than anything)
        for (int i = 0; i < n_in_use; ++i) used[i] = (int16_t)(n_in_use - i);     // symbol 0 in front, as the list begins
        int now = n_in_use;
        const int sweep = (n_in_use + 7) / 8;
        int64_t zrun = 0;
        auto flush_zeros = [&]() {
            if (zrun == 0) return;
            --zrun;
            for (;;) {                                          // bijective base 2: RUNA = 1, RUNB = 2
                const uint16_t s = (uint16_t)(zrun & 1);
                mtfv.push_back(s);
                mtf_freq[s]++;
                if (zrun < 2) break;
                zrun = (zrun - 2) / 2;
            }
            zrun = 0;
        };
        for (int32_t i = 0; i < nblock; ++i) {
            const uint8_t c = unseq_to_seq[last[(size_t)i]];
            const int mine = used[c];
            if (mine == now) { ++zrun; continue; }              // the front of the list
            flush_zeros();
            const __m128i ref = _mm_set1_epi16((short)mine);
            __m128i acc = _mm_setzero_si128();
            for (int q = 0; q < sweep; ++q)
                acc = _mm_sub_epi16(acc, _mm_cmpgt_epi16(_mm_load_si128(reinterpret_cast<const __m128i *>(used) + q), ref));
            acc = _mm_add_epi16(acc, _mm_srli_si128(acc, 8));
            acc = _mm_add_epi16(acc, _mm_srli_si128(acc, 4));
            acc = _mm_add_epi16(acc, _mm_srli_si128(acc, 2));
            const int j = _mm_cvtsi128_si32(acc) & 0xffff;      // symbols in front of c
            if (now == 32767) {
                // the times are running out: renumber them 1 .. n_in_use in their order (c's is set below)
                int16_t fresh[256];
                for (int a = 0; a < n_in_use; ++a) {
                    int before = 0;
                    for (int b2 = 0; b2 < n_in_use; ++b2) before += used[b2] < used[a];
                    fresh[a] = (int16_t)(before + 1);
                }
                for (int a = 0; a < n_in_use; ++a) used[a] = fresh[a];
                now = n_in_use;
            }
            used[c] = (int16_t)++now;
            mtfv.push_back((uint16_t)(j + 1));
            mtf_freq[j + 1]++;
        }
        flush_zeros();
        mtfv.push_back((uint16_t)eob);
        mtf_freq[eob]++;
    }
    const int32_t n_mtf = (int32_t)mtfv.size();
    lap(2);
    // ---- coding tables: initial split by frequency, 4 refinement rounds over groups of 50 symbols ----
    const int n_groups = n_mtf < 200 ? 2 : n_mtf < 600 ? 3 : n_mtf < 1200 ? 4 : n_mtf < 2400 ? 5 : 6;
    uint8_t len[kMaxGroups][kMaxAlpha];
    {
        int32_t rem = n_mtf;
        int gs = 0;
        for (int part = n_groups; part > 0; --part) {
            const int32_t target = rem / part;
            int ge = gs - 1;
            int32_t acc = 0;
            while (acc < target && ge < alpha - 1) { ++ge; acc += mtf_freq[ge]; }
            if (ge > gs && part != n_groups && part != 1 && ((n_groups - part) % 2 == 1)) { acc -= mtf_freq[ge]; --ge; }
            for (int v = 0; v < alpha; ++v) len[part - 1][v] = (v >= gs && v <= ge) ? 0 : 15;
            gs = ge + 1;
            rem -= acc;
        }
    }
    const int32_t n_sel = (n_mtf + kGroupSize - 1) / kGroupSize;
    std::vector<uint8_t> selector((size_t)n_sel);
    std::vector<int32_t> rfreq((size_t)kMaxGroups * kMaxAlpha);
    static_assert(kMaxGroups * 10 <= 64 && kGroupSize * 20 < 1024, "six 10-bit cost fields in one word");
    std::vector<uint64_t> packed((size_t)alpha);
    for (int iter = 0; iter < 4; ++iter) {
        std::fill(rfreq.begin(), rfreq.end(), 0);
        // (the cost of a group of 50 symbols under all tables at once: their code lengths side by side in one word)
        for (int v = 0; v < alpha; ++v) {
            uint64_t w = 0;
            for (int t = 0; t < n_groups; ++t) w |= (uint64_t)len[t][v] << (10 * t);
            packed[(size_t)v] = w;
        }
        for (int32_t g = 0; g < n_sel; ++g) {
            const int32_t a = g * kGroupSize, b = std::min<int32_t>(a + kGroupSize, n_mtf);
            int32_t best_cost = 0x7fffffff;
            int best = 0;
            uint64_t costs = 0;
            for (int32_t i = a; i < b; ++i) costs += packed[mtfv[(size_t)i]];
            for (int t = 0; t < n_groups; ++t) {
                const int32_t cost = (int32_t)((costs >> (10 * t)) & 1023u);
                if (cost < best_cost) { best_cost = cost; best = t; }
            }
            selector[(size_t)g] = (uint8_t)best;
            for (int32_t i = a; i < b; ++i) rfreq[(size_t)best * kMaxAlpha + mtfv[(size_t)i]]++;
        }
        for (int t = 0; t < n_groups; ++t) make_code_lengths(len[t], &rfreq[(size_t)t * kMaxAlpha], alpha, kEncCodeLen);
    }
    // canonical codes
    uint32_t code[kMaxGroups][kMaxAlpha];
    for (int t = 0; t < n_groups; ++t) {
        int mn = 32, mx = 0;
        for (int i = 0; i < alpha; ++i) { mn = std::min<int>(mn, len[t][i]); mx = std::max<int>(mx, len[t][i]); }
        uint32_t c = 0;
        for (int l = mn; l <= mx; ++l) {
            for (int i = 0; i < alpha; ++i) if (len[t][i] == l) code[t][i] = c++;
            c <<= 1;
        }
    }
    lap(3);
    // ---- the block ----
    bw.bits(24, (uint32_t)(kBlockMagic >> 24));
    bw.bits(24, (uint32_t)(kBlockMagic & 0xffffff));
    bw.bits(32, crc);
    bw.bits(1, 0);
    bw.bits(24, (uint32_t)orig_ptr);
    {
        uint32_t used16 = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j)
                if (in_use[i * 16 + j]) used16 |= 0x8000u >> i;
        bw.bits(16, used16);
        for (int i = 0; i < 16; ++i) {
            if (!(used16 & (0x8000u >> i))) continue;
            uint32_t m16 = 0;
            for (int j = 0; j < 16; ++j) if (in_use[i * 16 + j]) m16 |= 0x8000u >> j;
            bw.bits(16, m16);
        }
    }
    bw.bits(3, (uint32_t)n_groups);
    bw.bits(15, (uint32_t)n_sel);
    {
        uint8_t pos[kMaxGroups];
        for (int i = 0; i < n_groups; ++i) pos[i] = (uint8_t)i;
        for (int32_t g = 0; g < n_sel; ++g) {
            const uint8_t v = selector[(size_t)g];
            int j = 0;
            while (pos[j] != v) ++j;
            for (int k = j; k > 0; --k) pos[k] = pos[k - 1];
            pos[0] = v;
            for (int k = 0; k < j; ++k) bw.bits(1, 1);
            bw.bits(1, 0);
        }
    }
    for (int t = 0; t < n_groups; ++t) {
        int curr = len[t][0];
        bw.bits(5, (uint32_t)curr);
        for (int i = 0; i < alpha; ++i) {
            while (curr < len[t][i]) { bw.bits(2, 2); ++curr; }      // 10: increment
            while (curr > len[t][i]) { bw.bits(2, 3); --curr; }      // 11: decrement
            bw.bits(1, 0);
        }
    }
    for (int32_t g = 0; g < n_sel; ++g) {
        const int t = selector[(size_t)g];
        const int32_t a = g * kGroupSize, b = std::min<int32_t>(a + kGroupSize, n_mtf);
        for (int32_t i = a; i < b; ++i) bw.bits(len[t][mtfv[(size_t)i]], code[t][mtfv[(size_t)i]]);
    }
    lap(4);
    if (trace)
        fprintf(stderr, "[dq] bzip2 block of %d bytes (%d symbols): transform %.2f ms, last column %.2f, move to front %.2f, tables %.2f, bits %.2f; "
                "done at %.3f ms of the clock\n", nblock, n_mtf, t_ph[0], t_ph[1], t_ph[2], t_ph[3], t_ph[4],
                std::chrono::duration<double, std::milli>(now().time_since_epoch()).count() - 1e3 * (double)(long long)(std::chrono::duration<double>(now().time_since_epoch()).count() / 100.0) * 100.0);
    return 0;
}

// The blocks of a stream -- Burrows-Wheeler transform through the sorter, MTF / Huffman -- are independent and go to up
// to 4 threads (the extra stream of two unrelated 4 MiB files is five blocks of random bytes: 250 ms of move-to-front
// on one thread); their bit strings are then appended in order.
// Process-wide budget of extra framing threads (block encoders of all streams of all concurrent Diff.Create calls):
// one per hardware thread.  acquire() grants 0 ... want of them without waiting.
inline std::atomic<int> &framing_threads_in_use()
{
    static std::atomic<int> v{0};
    return v;
}
inline int framing_threads_acquire(int want)
{
    if (want <= 0) return 0;
    const int cap = (int)std::max(2u, std::thread::hardware_concurrency());
    std::atomic<int> &u = framing_threads_in_use();
    int cur = u.load();
    for (;;) {
        const int grant = std::min(want, std::max(0, cap - cur));
        if (grant == 0) return 0;
        if (u.compare_exchange_weak(cur, cur + grant)) return grant;
    }
}
inline void framing_threads_release(int n) { if (n > 0) framing_threads_in_use().fetch_sub(n); }

constexpr size_t kBlockSpan = (size_t)4 << 20;              // bytes of a stream one block covers at most (StreamEncoder)

// One stream, encoded while its bytes are still being produced.  feed() follows the producer with the run-length
// pre-pass and the block CRCs (which fix the block boundaries exactly as one sweep over the finished stream would: a
// run is only taken once the 255 bytes it may cover are known); a block that fills up goes to an encoder thread of
// its own, if the process-wide budget has one, while the pre-pass carries on; finish() encodes what is left -- on
// up to 8 threads -- and appends the bit strings in order.  The diff stream of two similar 16 MiB files is
// 16 MiB of mostly zeros: its pre-pass and CRC (3 + 2..8 ms) now run beside the device's anchor search instead of
// behind it, and the 900 KB blocks of the extra stream of unrelated files are encoded while the search goes on.
// One thread calls feed() / finish(); the sorter must be callable from several threads at once.
class StreamEncoder {
public:
    // span_max: a block also ends once it covers this many bytes of the stream.  (libbz2 cuts by coded size alone, which
    // makes the 16 MiB of mostly zeros of a diff stream ONE block of 425 KB that can only be transformed when the stream
    // is over -- 4.6 ms behind the scan of a Diff.Create that takes 12; blocks of at most 4 MiB of input are encoded
    // while the stream still grows, a quarter is left behind it, and the stream is ~1 % longer for their tables.  Which
    // blocks a stream is cut into depends on its bytes alone, never on how it was fed.)
    explicit StreamEncoder(DoubledSorter sorter_, int level_ = 9, size_t span_max_ = kBlockSpan)
        : sorter(std::move(sorter_)), level(level_), block_max((size_t)level_ * 100000 - 19), span_max(span_max_ ? span_max_ : (size_t)-1) {}
    StreamEncoder(const StreamEncoder &) = delete;
    StreamEncoder &operator=(const StreamEncoder &) = delete;
    ~StreamEncoder() { join_all(); }

    // src[0, upto) is final and will not move; `last`: the stream ends at upto.  (src and the earlier bytes are the
    // same from call to call.)
    void feed(const uint8_t *src, size_t upto, bool last)
    {
        // a run may reach 255 bytes ahead: positions whose run could still grow wait for the next call
        const size_t stop = last ? upto : (upto > 255 ? upto - 255 : 0);
        while (pos < stop) {
            if (!open) {
                blocks.emplace_back();
                blocks.back().rle.reserve(std::min(block_max, (last ? upto - pos : block_max)) + 8);
                open = true;
                crc = 0xffffffffu;
                crc_done = pos;
                block_from = pos;
            }
            std::vector<uint8_t> &blk = blocks.back().rle;
            size_t i = pos;
            // run-length pre-pass: a run of 4..255 equal bytes becomes 4 bytes + (length - 4)
            while (i < stop && blk.size() + 5 <= block_max && i - block_from < span_max) {
                // 16 bytes none of which equals its neighbour (15 of 16 such stretches of random or compressed data) are
                // 16 runs of one: copied as they are, while each of them would have passed the tests above on its own
                // and the byte behind them is known (4 MiB of random bytes: 18 ms of pre-pass a byte at a time)
                while (i + 17 <= upto && i + 16 <= stop && blk.size() + 16 + 5 <= block_max && i + 15 - block_from < span_max) {
                    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
                    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i + 1));
                    if (_mm_movemask_epi8(_mm_cmpeq_epi8(a, b))) break;
                    const size_t at = blk.size();
                    blk.resize(at + 16);
                    _mm_storeu_si128(reinterpret_cast<__m128i *>(blk.data() + at), a);
                    i += 16;
                }
                if (!(i < stop && blk.size() + 5 <= block_max && i - block_from < span_max)) break;
                const uint8_t c = src[i];
                size_t run = 1;
                const size_t lim = upto - i < 255 ? upto - i : 255;
                if (lim >= 16 && src[i + 1] == c) {
                    // a run has begun: extend it 8 bytes at a time
                    const uint64_t pat = 0x0101010101010101ull * c;
                    while (run + 8 <= lim) {
                        uint64_t v;
                        memcpy(&v, src + i + run, 8);
                        if (v != pat) break;
                        run += 8;
                    }
                }
                while (run < lim && src[i + run] == c) ++run;
                if (run >= 4) {
                    blk.insert(blk.end(), 4, c);
                    blk.push_back((uint8_t)(run - 4));
                } else {
                    blk.insert(blk.end(), run, c);
                }
                i += run;
                if (run == 255) {
                    // a long stretch of one byte: the runs of 255 that follow are counted 8 bytes a step and written
                    // as the five bytes each of them becomes, while they start in front of `stop` and fit the block
                    const uint64_t pat = 0x0101010101010101ull * c;
                    size_t z = 0;                              // further bytes equal to c from i on
                    while (i + z + 8 <= upto) {
                        uint64_t v;
                        memcpy(&v, src + i + z, 8);
                        if (v != pat) break;
                        z += 8;
                    }
                    size_t more = z / 255;
                    if (i < stop) more = std::min(more, (stop - i + 254) / 255); else more = 0;
                    more = std::min(more, (block_max - blk.size()) / 5);
                    if (i - block_from < span_max) more = std::min(more, (span_max - (i - block_from) + 254) / 255); else more = 0;
                    if (more > 0) {
                        const size_t at = blk.size();
                        blk.resize(at + 5 * more);
                        uint8_t *o = blk.data() + at;
                        for (size_t k = 0; k < more; ++k, o += 5) { o[0] = o[1] = o[2] = o[3] = c; o[4] = 251; }
                        i += 255 * more;
                    }
                }
            }
            pos = i;
            crc = crc_update_mt(crc, src + crc_done, pos - crc_done);    // of the block's input bytes (4 threads from 2 MiB a call)
            crc_done = pos;
            if (blk.size() + 5 > block_max || pos - block_from >= span_max) close_block(/*more_to_come=*/true);
        }
        if (last && open) close_block(false);
    }

    // after feed(.., last = true): the whole stream "BZh<level>" ... end magic, appended to out
    int finish(std::vector<uint8_t> &out)
    {
        if (open) close_block(false);
        // what no thread has claimed yet: the caller and up to 7 more threads take the blocks in turn
        size_t pending = 0;
        for (Block &b : blocks) pending += b.claimed.load() == 0;
        if (pending >= 2) {
            size_t extra = std::min<size_t>(pending, 8) - 1;
            const int granted = framing_threads_acquire((int)extra);
            std::vector<std::thread> ts;
            for (int t = 0; t < granted; ++t) {
                try { ts.emplace_back([this] { encode_pending(); }); } catch (...) { break; }
            }
            encode_pending();
            for (std::thread &t : ts) t.join();
            framing_threads_release(granted);
        } else {
            encode_pending();
        }
        join_all();
        BitWriter bw(out);
        bw.bits(8, 'B'); bw.bits(8, 'Z'); bw.bits(8, 'h'); bw.bits(8, (uint32_t)('0' + level));
        uint32_t combined = 0;
        for (Block &b : blocks) {
            if (b.rc != 0) return b.rc;
            bw.append(b.bytes.data(), b.nbits);
            combined = ((combined << 1) | (combined >> 31)) ^ b.crc;
        }
        bw.bits(24, (uint32_t)(kEndMagic >> 24));
        bw.bits(24, (uint32_t)(kEndMagic & 0xffffff));
        bw.bits(32, combined);
        bw.flush();
        return 0;
    }

    size_t consumed() const { return pos; }

    // (tests: the run-length coded blocks and their CRCs as the pre-pass cut them, before anything is encoded --
    // constructed with hold = true, so that no block leaves for an encoder thread)
    void hold_blocks() { hold = true; }
    void copy_blocks(std::vector<uint8_t> &rle, std::vector<uint32_t> &lens, std::vector<uint32_t> &crcs)
    {
        if (open) close_block(false);
        for (Block &b : blocks) {
            rle.insert(rle.end(), b.rle.begin(), b.rle.end());
            lens.push_back((uint32_t)b.rle.size());
            crcs.push_back(b.crc);
        }
    }

private:
    struct Block {
        std::vector<uint8_t> rle;        // run-length coded input of the block
        uint32_t crc = 0;
        std::vector<uint8_t> bytes;      // its bit string ...
        uint64_t nbits = 0;              // ... and how many bits of it count
        int rc = 0;
        std::atomic<int> claimed{0};     // an encoder has taken it
    };
    struct Helper {
        std::thread t;
    };

    // (nothing may leave a worker thread as an exception -- that would be std::terminate, not an error code: an
    // allocation that fails inside a block's encoder becomes that block's rc)
    void encode(Block &b)
    {
        try {
            BitWriter w(b.bytes);
            b.rc = compress_block(w, b.rle, b.crc, sorter);
            b.nbits = w.total;
            w.flush();
            std::vector<uint8_t>().swap(b.rle);
        } catch (...) {
            b.rc = -3;
        }
    }
    void encode_pending()
    {
        for (Block &b : blocks) {
            int expect = 0;
            if (b.claimed.compare_exchange_strong(expect, 1)) encode(b);
        }
    }
    void close_block(bool more_to_come)
    {
        Block &b = blocks.back();
        b.crc = ~crc;
        open = false;
        if (!more_to_come || hold) return;   // (the last block is finish()'s: the caller is about to wait for it anyway)
        // full while the stream is still growing: encode it beside the pre-pass if a framing thread is to be had
        if (framing_threads_acquire(1) != 1) return;
        int expect = 0;
        if (!b.claimed.compare_exchange_strong(expect, 1)) { framing_threads_release(1); return; }
        try {
            helpers.emplace_back();
            helpers.back().t = std::thread([this, &b] { encode(b); framing_threads_release(1); });
        } catch (...) {
            if (!helpers.empty() && !helpers.back().t.joinable()) helpers.pop_back();
            b.claimed.store(0);
            framing_threads_release(1);
        }
    }
    void join_all()
    {
        for (Helper &h : helpers) if (h.t.joinable()) h.t.join();
        helpers.clear();
    }

    DoubledSorter sorter;
    int level;
    size_t block_max;
    std::deque<Block> blocks;            // (a deque: blocks stay where they are while their encoders run)
    std::deque<Helper> helpers;
    size_t span_max;
    size_t pos = 0, crc_done = 0, block_from = 0;
    uint32_t crc = 0;
    bool open = false;
    bool hold = false;
};

// level 1..9: blocks of level * 100000 - 19 run-length coded bytes (libbz2's limit).
inline int bz2_compress(const uint8_t *src, size_t n, std::vector<uint8_t> &out, const DoubledSorter &sorter, int level = 9,
                        size_t span_max = kBlockSpan)
{
    StreamEncoder enc(sorter, level, span_max);
    enc.feed(src, n, true);
    return enc.finish(out);
}

}  // namespace bz2
}  // namespace dq
