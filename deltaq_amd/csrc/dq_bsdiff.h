// dq_bsdiff.h -- the BSDIFF40 container around the hot path: Diff.Create and Patch.Apply, natively (host code
// driving the device kernels).  SURVEY.md section 8(f) row 3.
//
// Reference: src/DeltaQ.BsDiff/Diff.cs:27-241 (Create: header :54-70, suffix sort :89-90, scan loop :100-232, the
//            three bzip2 streams :85-87 / :235-245, header rewrite :247-252), Patch.cs:52-168 (Apply),
//            SpanExtensions.cs:7-44 (packed longs), Constants.cs:5-12 (layout, "BSDIFF40").
//
//   old file --H2D--> suffix array (the HIP sorter, stays on the device) --> match search kernel, asked for
//   WINDOWS of scan positions by the sequential scan loop on the host --> control triples, diff and extra bytes
//   --> three bzip2 streams (dq_bz2.h; the Burrows-Wheeler transform of every block is one more run of the HIP
//   sorter) --> header + streams.
//
// The scan loop consumes Search only through (pos, len) (Diff.cs:106), so every decision of it is kept as the
// reference has it (scan_loop() below) and it reads from a window of answers the device filled speculatively for
// the positions ahead of `scan`:
// in a region where old and new differ the loop advances byte by byte and uses every answer; after a match it
// jumps by `len`, and a jump out of the window simply starts the next window at the new position.  A position
// whose comparison ran into the cap (inside a long match) comes back undecided and is asked again, exactly, as
// the first position of a new window.
// The patch is therefore the one the reference's loop produces, byte for byte in its raw streams
// (tests/test_gpu_bsdiff.py compares them with the oracle's restatement); the bzip2 framing is a valid encoding of
// them, not necessarily SharpZipLib's bytes (which the reference does not pin either).
#pragma once
#include <chrono>
#include <stdint.h>

#include <atomic>
#include <cstring>
#include <vector>

namespace dq {
namespace bsdiff {

constexpr int kHeaderSize = 32;                                   // Constants.cs:7
constexpr int64_t kSignature = 0x3034464649445342ll;              // "BSDIFF40", Constants.cs:14

// SpanExtensions.cs:7-29: sign-magnitude, little endian, bit 7 of byte 7 = sign
inline void write_packed_long(uint8_t *p, int64_t y)
{
    uint64_t u;
    uint8_t sign = 0;
    if (y < 0) { u = (uint64_t)(-(y + 1)) + 1; sign = 0x80; } else u = (uint64_t)y;
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(u >> (8 * i));
    p[7] = (uint8_t)((p[7] & 0x7f) | sign);
}

// SpanExtensions.cs:31-44
inline int64_t read_packed_long(const uint8_t *p)
{
    int64_t y = p[7] & 0x7f;
    for (int i = 6; i >= 0; --i) y = (y << 8) + p[i];
    return (p[7] & 0x80) ? -y : y;
}

// ---- byte-run helpers of the scan loop (8 bytes a step; the loop walks every byte of both files several times) ----
inline uint64_t load_u64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

// number of i < len with a[i] == b[i]
inline int64_t count_equal(const uint8_t *a, const uint8_t *b, int64_t len)
{
    int64_t i = 0, c = 0;
    constexpr uint64_t k7f = 0x7f7f7f7f7f7f7f7full;
    for (; i + 8 <= len; i += 8) {
        const uint64_t x = load_u64(a + i) ^ load_u64(b + i);
        const uint64_t t = ~(((x & k7f) + k7f) | x | k7f);        // 0x80 in every byte of x that is zero
        c += __builtin_popcountll(t);
    }
    for (; i < len; ++i) c += a[i] == b[i];
    return c;
}

// number of zero bytes of x (x = a ^ b: the bytes two words agree in)
inline int equal_bytes(uint64_t x)
{
    constexpr uint64_t k7f = 0x7f7f7f7f7f7f7f7full;
    return __builtin_popcountll(~(((x & k7f) + k7f) | x | k7f));
}

// length of the common prefix of a[0..len) and b[0..len)
inline int64_t common_prefix(const uint8_t *a, const uint8_t *b, int64_t len)
{
    int64_t i = 0;
    // (32 bytes a step while they are equal: between similar files the extensions are runs of kilobytes)
    for (; i + 32 <= len; i += 32) {
        const uint64_t x = (load_u64(a + i) ^ load_u64(b + i)) | (load_u64(a + i + 8) ^ load_u64(b + i + 8)) |
                           (load_u64(a + i + 16) ^ load_u64(b + i + 16)) | (load_u64(a + i + 24) ^ load_u64(b + i + 24));
        if (x) break;
    }
    for (; i + 8 <= len; i += 8) {
        const uint64_t x = load_u64(a + i) ^ load_u64(b + i);
        if (x) return i + (__builtin_ctzll(x) >> 3);              // (little endian: the first differing byte is the lowest)
    }
    for (; i < len && a[i] == b[i]; ++i) {}
    return i;
}

// number of k in 1..len with a[-k] == b[-k] for all 1..k (common suffix of the bytes before a and b)
inline int64_t common_suffix(const uint8_t *a, const uint8_t *b, int64_t len)
{
    int64_t k = 0;
    for (; k + 32 <= len; k += 32) {
        const uint64_t x = (load_u64(a - k - 8) ^ load_u64(b - k - 8)) | (load_u64(a - k - 16) ^ load_u64(b - k - 16)) |
                           (load_u64(a - k - 24) ^ load_u64(b - k - 24)) | (load_u64(a - k - 32) ^ load_u64(b - k - 32));
        if (x) break;
    }
    for (; k + 8 <= len; k += 8) {
        const uint64_t x = load_u64(a - k - 8) ^ load_u64(b - k - 8);
        if (x) return k + (__builtin_clzll(x) >> 3);              // the last byte of the word is the nearest one
    }
    for (; k < len && a[-k - 1] == b[-k - 1]; ++k) {}
    return k;
}

struct RawStreams {
    std::vector<uint8_t> ctrl, diff, extra;                       // ctrl: 24 bytes (three packed longs) per triple
    int64_t searches = 0, windows = 0, exact = 0;
};

// The greedy alignment of bsdiff as the reference runs it (Diff.cs:91-232), in its three steps per control triple:
//   1. anchor   from the end of the last match, Search every position until it returns a match that is not just
//               the previous alignment carried on (more than 8 bytes better than what old[. + shift] already
//               gives), or one that the previous alignment explains completely                    (:104-125)
//   2. extend   forwards from the previous anchor and backwards from the new one, each to the length with the
//               best 2 * matches - length; where the two extensions overlap, cut at the best split  (:129-191)
//   3. emit     the forward extension as diff bytes (new - old), what lies between the extensions as extra
//               bytes, and the triple (diff length, extra length, seek in old)                        (:196-223)
// `search(scan, &pos, &len)` stands for Search(I, old, new[scan..], 0, n, out pos): 0 or an error code.  Every
// decision is the reference's, so the triples and both byte streams are the reference's (the tests compare them
// with the oracle's restatement of the same lines).
// Steps 2 and 3 for one anchor: the state they carry from triple to triple is `prev`.
struct TripleEmitter {
    const uint8_t *old;
    int64_t n;
    const uint8_t *nw;
    int64_t m;
    RawStreams &out;
    struct Anchor { int64_t at = 0, in_old = 0; };          // a matched position of new and where it lies in old
    Anchor prev;                                             // end of the last emitted forward extension
    // if set: the lengths of out.diff / out.extra that are final, published after every triple for a thread that
    // frames the streams while they grow (the vectors must have their full capacity reserved: they may not move)
    std::atomic<size_t> *progress = nullptr;
    double *phase_ms = nullptr;                              // (DQ_TRACE) [3]: time in the extensions, in the diff bytes, in the rest

    TripleEmitter(const uint8_t *old_, int64_t n_, const uint8_t *nw_, int64_t m_, RawStreams &out_)
        : old(old_), n(n_), nw(nw_), m(m_), out(out_) {}

    void emit_packed(int64_t v)
    {
        uint8_t b[8];
        write_packed_long(b, v);
        out.ctrl.insert(out.ctrl.end(), b, b + 8);
    }

    // the anchor the loop broke on: position `cursor` of new matched at `hit_pos` of old (cursor == m: the end of new)
    void take(int64_t cursor, int64_t hit_pos)
    {
        // ---- 2. extensions ----
        const auto t_a = phase_ms ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        int64_t fwd = 0;                                     // forward from prev, under prev's alignment
        int64_t equal_front = 0;                             // bytes at the front of the extension known to be equal
        for (int64_t i = 0, good = 0, best = 0; prev.at + i < cursor && prev.in_old + i < n;) {
            if (good == best && i == fwd) {
                // standing on the best prefix so far: every further equal byte makes a better one, so a run of them
                // is taken in one step (the files are mostly such runs)
                const int64_t room = (cursor - prev.at - i) < (n - prev.in_old - i) ? (cursor - prev.at - i) : (n - prev.in_old - i);
                const int64_t run = common_prefix(old + prev.in_old + i, nw + prev.at + i, room);
                if (i == 0) equal_front = run;
                good += run;
                i += run;
                best = good;
                fwd = i;
                if (run == room) break;
            }
            // (a long stretch the two files do not share -- new data, unrelated files: eight positions at once where none of
            // them can be a new best; k equal bytes among them lift 2 * good - i by at most k over its value in front of
            // them.  4 MiB of unrelated bytes: 4.2 -> ~1 ms)
            while (prev.at + i + 8 <= cursor && prev.in_old + i + 8 <= n) {
                const int k = equal_bytes(load_u64(old + prev.in_old + i) ^ load_u64(nw + prev.at + i));
                if (k > 0 && 2 * good - i + k > 2 * best - fwd) break;
                good += k;
                i += 8;
            }
            if (!(prev.at + i < cursor && prev.in_old + i < n)) break;
            good += old[prev.in_old + i] == nw[prev.at + i];
            ++i;
            if (2 * good - i > 2 * best - fwd) { best = good; fwd = i; }
        }
        int64_t back = 0;                                    // backward from the new anchor, under its alignment
        if (cursor < m) {
            for (int64_t i = 1, good = 0, best = 0; cursor >= prev.at + i && hit_pos >= i; ++i) {
                if (good == best && back == i - 1) {           // as above, towards the front
                    const int64_t room = ((cursor - prev.at) < hit_pos ? (cursor - prev.at) : hit_pos) - (i - 1);
                    const int64_t run = common_suffix(old + hit_pos - (i - 1), nw + cursor - (i - 1), room);
                    good += run;
                    i += run;
                    best = good;
                    back = i - 1;
                    if (run == room) break;
                }
                // The walk goes all the way back to the previous anchor (Diff.cs:144-152) -- 80 kB a triple between files
                // that differ in 200 places, under an alignment that is wrong there: 18 ms of single bytes per 16 MiB
                // pair.  Eight positions at once where none of them can be a new best: k equal bytes among them lift
                // 2 * good - i by at most k + 1 over its value in front of them.
                while (cursor >= prev.at + i + 7 && hit_pos >= i + 7) {
                    const int k = equal_bytes(load_u64(old + hit_pos - i - 7) ^ load_u64(nw + cursor - i - 7));
                    if (2 * good - i + (k > 0 ? k + 1 : 0) > 2 * best - back) break;
                    good += k;
                    i += 8;
                }
                if (!(cursor >= prev.at + i && hit_pos >= i)) break;
                good += old[hit_pos - i] == nw[cursor - i];
                if (2 * good - i > 2 * best - back) { best = good; back = i; }
            }
        }
        const int64_t clash = (prev.at + fwd) - (cursor - back);
        if (clash > 0) {                                     // both claim `clash` bytes: give each side its better part
            int64_t balance = 0, best = 0, cut = 0;
            for (int64_t i = 0; i < clash; ++i) {
                balance += nw[prev.at + fwd - clash + i] == old[prev.in_old + fwd - clash + i];
                balance -= nw[cursor - back + i] == old[hit_pos - back + i];
                if (balance > best) { best = balance; cut = i + 1; }
            }
            fwd += cut - clash;
            back -= cut;
        }

        // ---- 3. one control triple ----
        const auto t_b = phase_ms ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        const size_t d0 = out.diff.size();
        out.diff.resize(d0 + (size_t)fwd);                   // (zeros)
        {
            // new - old, byte by byte -- but the forward extension is mostly equal bytes (that is what made it the
            // extension): whole words of them are left as the zeros they already are
            uint8_t *dst = out.diff.data() + d0;
            const uint8_t *a = nw + prev.at, *b = old + prev.in_old;
            // (the equal bytes the extension began with -- between similar files nearly all of it -- are not read again)
            int64_t i = (equal_front < fwd ? equal_front : fwd) & ~(int64_t)7;
            for (; i + 8 <= fwd; i += 8) {
                if (load_u64(a + i) == load_u64(b + i)) continue;
                for (int q = 0; q < 8; ++q) dst[i + q] = (uint8_t)(a[i + q] - b[i + q]);
            }
            for (; i < fwd; ++i) dst[i] = (uint8_t)(a[i] - b[i]);
        }
        const auto t_c = phase_ms ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        const int64_t gap = (cursor - back) - (prev.at + fwd);
        if (gap > 0) out.extra.insert(out.extra.end(), nw + prev.at + fwd, nw + prev.at + fwd + gap);
        emit_packed(fwd);
        emit_packed(gap);
        emit_packed((hit_pos - back) - (prev.in_old + fwd));
        prev.at = cursor - back;
        prev.in_old = hit_pos - back;
        if (progress) {
            progress[0].store(out.diff.size(), std::memory_order_release);
            progress[1].store(out.extra.size(), std::memory_order_release);
        }
        if (phase_ms) {
            const auto t_d = std::chrono::steady_clock::now();
            phase_ms[0] += std::chrono::duration<double, std::milli>(t_b - t_a).count();
            phase_ms[1] += std::chrono::duration<double, std::milli>(t_c - t_b).count();
            phase_ms[2] += std::chrono::duration<double, std::milli>(t_d - t_c).count();
        }
    }
};

template <typename SearchFn>
int scan_loop(const uint8_t *old, int64_t n, const uint8_t *nw, int64_t m, SearchFn &&search, RawStreams &out)
{
    TripleEmitter em(old, n, nw, m, out);
    int64_t shift = 0;                                       // old - new offset of the previous alignment
    int64_t cursor = 0, hit_pos = 0, hit_len = 0;            // scan position, last Search answer

    auto agrees = [&](int64_t i) { return i + shift < n && old[i + shift] == nw[i]; };     // previous alignment still right at i?

    while (cursor < m) {
        // ---- 1. next anchor ----
        int64_t carried = 0;                                 // bytes of [.., cursor + hit_len) the previous alignment gets right
        int64_t counted = cursor += hit_len;                 // ... counted up to here
        for (; cursor < m; ++cursor) {
            const int rc = search(cursor, &hit_pos, &hit_len);
            if (rc != 0) return rc;
            ++out.searches;
            if (counted < cursor + hit_len) {                // bytes of the new match the previous alignment also gets right
                const int64_t end = cursor + hit_len, upto = end < n - shift ? end : n - shift;
                if (upto > counted) carried += count_equal(old + counted + shift, nw + counted, upto - counted);
                counted = end;
            }
            if ((hit_len == carried && hit_len != 0) || hit_len > carried + 8) break;
            carried -= agrees(cursor);
        }
        if (hit_len == carried && cursor != m) continue;     // the old alignment explains it: keep scanning behind it
        em.take(cursor, hit_pos);
        shift = hit_pos - cursor;
    }
    return 0;
}

// The same streams from the ANCHORS alone: step 1 runs on the device (dq_anchor_scan.h: the whole anchor search of
// Diff.cs:100-125 without a host round trip per window) and hands over, per control triple, the position the loop
// broke on and where its match lies in old -- (cursor, hit_pos) pairs in order, the last one with cursor == m.
// The emitter keeps its state between calls, so the pairs may arrive in several batches.
inline void scan_from_anchors(TripleEmitter &em, const int64_t *pairs, int64_t count)
{
    for (int64_t k = 0; k < count; ++k) em.take(pairs[2 * k], pairs[2 * k + 1]);
}

// Patch.cs:95-168 on the three decoded streams.  Returns 0, or -1 for what the reference reports as "Corrupt patch".
// Every bound is checked in a form that cannot overflow: the three numbers of a triple are attacker-controlled
// 63-bit values (CVE-2014-9862 is this bug class in the original bspatch).
inline int apply_streams(const uint8_t *old, int64_t n, const std::vector<uint8_t> &ctrl, const std::vector<uint8_t> &diff,
                         const std::vector<uint8_t> &extra, int64_t newsize, uint8_t *out)
{
    int64_t outpos = 0, oldpos = 0;
    size_t cpos = 0, dpos = 0, epos = 0;
    while (outpos < newsize) {
        if (ctrl.size() - cpos < 24) return -1;
        const int64_t add = read_packed_long(&ctrl[cpos]), copy = read_packed_long(&ctrl[cpos + 8]),
                      seek = read_packed_long(&ctrl[cpos + 16]);
        cpos += 24;
        if (add < 0 || copy < 0 || add > newsize - outpos) return -1;                // :131 sanity-check
        // short reads (:139-140).  With add == 0 nothing is read, wherever the old-file position stands.
        if ((uint64_t)add > diff.size() - dpos || (add > 0 && (oldpos > n || add > n - oldpos))) return -1;
        for (int64_t i = 0; i < add; i++) out[outpos + i] = (uint8_t)(diff[dpos + (size_t)i] + old[oldpos + i]);
        outpos += add; dpos += (size_t)add; oldpos += add;
        if (copy > newsize - outpos || (uint64_t)copy > extra.size() - epos) return -1;          // :153
        if (copy > 0) memcpy(out + outpos, &extra[epos], (size_t)copy);
        outpos += copy; epos += (size_t)copy;
        // :165 -- Stream.Seek to a negative position throws in the reference; a wrapped sum is never a valid position
        if (__builtin_add_overflow(oldpos, seek, &oldpos) || oldpos < 0) return -1;
    }
    return 0;
}

struct Header {
    int64_t ctrl_len = 0, diff_len = 0, new_size = 0;
};

// Patch.cs:52-93.  0, or -1 ("Corrupt patch").
inline int parse_header(const uint8_t *patch, int64_t plen, Header *h)
{
    if (plen < kHeaderSize) return -1;
    if (read_packed_long(patch) != kSignature) return -1;
    h->ctrl_len = read_packed_long(patch + 8);
    h->diff_len = read_packed_long(patch + 16);
    h->new_size = read_packed_long(patch + 24);
    if (h->ctrl_len < 0 || h->diff_len < 0 || h->new_size < 0) return -1;
    // (no sums of the two lengths: each is a 63-bit value taken from the file)
    if (h->ctrl_len > plen - kHeaderSize || h->diff_len > plen - kHeaderSize - h->ctrl_len) return -1;
    return 0;
}

}  // namespace bsdiff
}  // namespace dq
