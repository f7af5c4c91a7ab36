// dq_alpha_code.h -- an order-preserving, length-limited prefix code for the bytes of a text (host code).
//
// Round 0 of a text-like input sorts 64-bit keys.  Eight raw bytes fill them; the same 64 bits hold about
// 64 / H0 characters when every byte is written as a codeword of an ALPHABETIC prefix code (codeword order =
// byte order, so comparing the concatenated codewords as bit strings is comparing the texts), which a skewed
// alphabet makes 12...14 characters instead of 8: far fewer suffixes are still tied after round 0, and those
// are what the doubling rounds pay for.  (Reference context: the path replaced is LibDivSufSort.Sort,
// src/DeltaQ.SuffixSorting.LibDivSufSort/LibDivSufSort.cs:12-29; nothing in the reference corresponds to
// this step -- its output, the suffix array, is unchanged by it.)
//
// Constraints on the code, all of them needed by the device side (dq_onesweep.h, coded_keys4()):
//   * codeword lengths in [kCodeMinLen, kCodeMaxLen] = [4, 8]: two suffixes with equal 64-bit keys share at
//     least 8 whole characters (the doubling rounds start from h = 8 exactly as with raw bytes), and 16
//     characters always fill the 64 bits (so a key never ends in padding inside the text);
//   * the smallest present byte gets the all-zero codeword and absent bytes read as zeros: the zero bytes
//     behind the end of the text stay the smallest possible continuation.
// Among the codes that satisfy them this one minimises the expected length: dynamic programme over
// (levels left, byte interval), split points bounded by Knuth's monotonicity inside the window the level's
// capacity allows, O(sigma^2) cells per level (0.1 ms for 73 symbols, ~1 ms for 256 on one host core).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <vector>

namespace dq {

constexpr int kCodeMinLen = 4;
constexpr int kCodeMaxLen = 8;

struct AlphaCode {
    uint16_t tab[256];      // codeword << 4 | length; absent bytes: 0 << 4 | 8
    double avg_len;         // expected bits per text byte
    int sigma;              // distinct bytes
};

inline bool build_alpha_code(const int64_t hist[256], AlphaCode *out)
{
    int sym[256], s = 0;
    for (int b = 0; b < 256; ++b) {
        out->tab[b] = (uint16_t)kCodeMaxLen;
        if (hist[b] > 0) sym[s++] = b;
    }
    out->sigma = s;
    out->avg_len = kCodeMaxLen;
    if (s == 0) return false;
    int64_t P[257];
    P[0] = 0;
    for (int i = 0; i < s; ++i) P[i + 1] = P[i] + hist[sym[i]];
    constexpr int64_t kInf = INT64_MAX / 4;
    constexpr int R = kCodeMaxLen;                                // levels available below the root
    constexpr int kDim = 257;
    // cost[i][j]: cheapest subtree over symbols [i, j) whose root has r levels left (depth R - r); a leaf above
    // depth kCodeMinLen is padded to it.  Two levels at a time (this one and the one below, the latter also
    // transposed: both operands of the split loop are then contiguous), splits for every level.  Only the cells a
    // level can use are touched: intervals of at most 2^r symbols.
    struct Tables {
        int64_t cur[kDim][kDim], prev[kDim][kDim], prev_t[kDim][kDim];
        uint8_t split[R + 1][kDim][kDim];                         // split point - i  (1 .. 255)
    };
    // (2.2 MB, allocated per call and left uninitialised: only the cells a level writes are ever read, so only those
    // pages are touched; nothing stays behind on the calling thread)
    std::unique_ptr<Tables> store(new Tables);
    Tables &t = *store;
    for (int r = 0; r <= R; ++r) {
        const int depth = R - r;
        const int pad = depth < kCodeMinLen ? kCodeMinLen - depth : 0;
        const int room = 1 << r;                                  // leaves a subtree of r levels can hold
        const int below = r > 0 ? 1 << (r - 1) : 0;               // ... and a child subtree
        for (int i = 0; i < s; ++i) t.cur[i][i + 1] = (int64_t)pad * hist[sym[i]];
        for (int len = 2; len <= s && len <= room; ++len) {
            for (int i = 0; i + len <= s; ++i) {
                const int j = i + len;
                // both children must fit `below` leaves: k - i <= below and j - k <= below
                int lo = j - below > i + 1 ? j - below : i + 1, hi = i + below < j - 1 ? i + below : j - 1;
                if (len > 2 && len - 1 <= room) {
                    // Knuth's bounds from the two intervals one symbol shorter (same level), where they exist
                    const int a = i + t.split[r][i][j - 1], b = i + 1 + t.split[r][i + 1][j];
                    if (a <= b && a >= lo && b <= hi) { lo = a; hi = b; }
                }
                int64_t best = kInf;
                int bk = 0;
                const int64_t *left = t.prev[i], *right = t.prev_t[j];
                for (int k = lo; k <= hi; ++k) {
                    const int64_t c = left[k] + right[k];
                    if (c < best) { best = c; bk = k; }
                }
                t.cur[i][j] = best + (P[j] - P[i]);
                t.split[r][i][j] = (uint8_t)(bk - i);
            }
        }
        if (r == R) break;
        // this level becomes the one below: only the cells the next level reads (intervals of <= 2^r symbols)
        for (int i = 0; i < s; ++i) {
            const int jmax = i + room < s ? i + room : s;
            for (int j = i + 1; j <= jmax; ++j) { t.prev[i][j] = t.cur[i][j]; t.prev_t[j][i] = t.cur[i][j]; }
        }
    }
    if (s > (1 << R)) return false;                               // (cannot happen: 256 = 2^8)
    // walk the tree: explicit stack of (r, i, j, prefix, depth)
    struct Node { int r, i, j; uint32_t prefix; int depth; };
    Node stack[2 * R + 4];
    int sp = 0;
    stack[sp++] = {R, 0, s, 0u, 0};
    int64_t bits = 0;
    while (sp > 0) {
        const Node nd = stack[--sp];
        if (nd.j - nd.i == 1) {
            const int len = nd.depth < kCodeMinLen ? kCodeMinLen : nd.depth;
            const uint32_t code = nd.prefix << (len - nd.depth);
            out->tab[sym[nd.i]] = (uint16_t)((code << 4) | (uint32_t)len);
            bits += (int64_t)len * hist[sym[nd.i]];
            continue;
        }
        const int k = nd.i + t.split[nd.r][nd.i][nd.j];
        stack[sp++] = {nd.r - 1, nd.i, k, nd.prefix << 1, nd.depth + 1};
        stack[sp++] = {nd.r - 1, k, nd.j, (nd.prefix << 1) | 1u, nd.depth + 1};
    }
    out->avg_len = (double)bits / (double)P[s];
    return true;
}

}  // namespace dq
