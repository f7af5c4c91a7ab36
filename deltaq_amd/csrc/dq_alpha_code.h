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
// (levels left, byte interval), split points bounded by Knuth's monotonicity (with a full scan as fallback
// where the bound would exclude every feasible split), O(sigma^2) cells per level.
#pragma once
#include <stddef.h>
#include <stdint.h>

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
    std::vector<int64_t> P((size_t)s + 1, 0);
    for (int i = 0; i < s; ++i) P[(size_t)i + 1] = P[(size_t)i] + hist[sym[i]];
    constexpr int64_t kInf = INT64_MAX / 4;
    const int R = kCodeMaxLen;                                    // levels available below the root
    const size_t dim = (size_t)s + 1;
    // cost[r][i][j]: cheapest subtree over symbols [i, j) whose root has r levels left (depth R - r);
    // a leaf above depth kCodeMinLen is padded to it
    std::vector<int64_t> cost((size_t)(R + 1) * dim * dim, kInf);
    std::vector<uint16_t> split((size_t)(R + 1) * dim * dim, 0);
    auto at = [&](int r, int i, int j) -> size_t { return ((size_t)r * dim + (size_t)i) * dim + (size_t)j; };
    for (int r = 0; r <= R; ++r) {
        const int depth = R - r;
        const int pad = depth < kCodeMinLen ? kCodeMinLen - depth : 0;
        for (int i = 0; i < s; ++i) cost[at(r, i, i + 1)] = (int64_t)pad * hist[sym[i]];
        if (r == 0) continue;
        const int64_t room = r >= 31 ? INT64_MAX : (1ll << r);    // leaves a subtree of r levels can hold
        for (int len = 2; len <= s; ++len) {
            if (len > room) break;
            for (int i = 0; i + len <= s; ++i) {
                const int j = i + len;
                int lo = i + 1, hi = j - 1;
                if (len > 2) {
                    const int a = split[at(r, i, j - 1)], b = split[at(r, i + 1, j)];
                    if (a > 0 && b > 0 && a <= b) { lo = a; hi = b < j - 1 ? b : j - 1; }
                }
                int64_t best = kInf;
                int bk = 0;
                for (int pass = 0; pass < 2 && bk == 0; ++pass) {
                    if (pass == 1) { lo = i + 1; hi = j - 1; }    // the bounded window held no feasible split
                    for (int k = lo; k <= hi; ++k) {
                        const int64_t a = cost[at(r - 1, i, k)], b = cost[at(r - 1, k, j)];
                        if (a >= kInf || b >= kInf) continue;
                        if (a + b < best) { best = a + b; bk = k; }
                    }
                }
                if (bk) {
                    cost[at(r, i, j)] = best + (P[(size_t)j] - P[(size_t)i]);
                    split[at(r, i, j)] = (uint16_t)bk;
                }
            }
        }
    }
    if (cost[at(R, 0, s)] >= kInf) return false;                  // (cannot happen: 256 <= 2^8)
    // walk the tree: explicit stack of (r, i, j, prefix, depth)
    struct Node { int r, i, j; uint32_t prefix; int depth; };
    std::vector<Node> stack;
    stack.push_back({R, 0, s, 0u, 0});
    int64_t bits = 0;
    while (!stack.empty()) {
        const Node nd = stack.back();
        stack.pop_back();
        if (nd.j - nd.i == 1) {
            const int len = nd.depth < kCodeMinLen ? kCodeMinLen : nd.depth;
            const uint32_t code = nd.prefix << (len - nd.depth);
            out->tab[sym[nd.i]] = (uint16_t)((code << 4) | (uint32_t)len);
            bits += (int64_t)len * hist[sym[nd.i]];
            continue;
        }
        const int k = split[at(nd.r, nd.i, nd.j)];
        stack.push_back({nd.r - 1, nd.i, k, nd.prefix << 1, nd.depth + 1});
        stack.push_back({nd.r - 1, k, nd.j, (nd.prefix << 1) | 1u, nd.depth + 1});
    }
    out->avg_len = (double)bits / (double)P[(size_t)s];
    return true;
}

}  // namespace dq
