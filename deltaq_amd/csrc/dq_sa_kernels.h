// dq_sa_kernels.h -- suffix-array specific kernels around the radix engine:
//   gather_key2      composite (rank, ISA[s + h]) keys for the next doubling round
//   gather_text_key  composite (rank, next bytes of text) keys for the sparse finishing rounds
//   small_group_finish   groups of <= 8 tied suffixes sorted by direct text comparison
//   sample_ties, isa_from_sa, isa_scatter
//
// Order contract (reference: LibDivSufSortTests.cs:43-59 -- unsigned bytes,
// a proper prefix sorts first): suffixes that run off the end of the text are
// resolved by gather_key2's "past the end" rule, never by the zero padding.
#pragma once
#include "dq_runs.h"
#include "dq_device_utils.h"

namespace dq {

// ---------------------------------------------------------------------------------
// gather_key2: composite[j] = (rank_j << kbits) | key2_j in place, where
//   key2 = ISA[s + h] + h          when s + h < n   (rank of the suffix h further on)
//        = n - 1 - s               otherwise        (ran off the end: the SHORTER suffix,
//                                                    i.e. the larger s, sorts first)
// Past-the-end values are < h, in-range values are >= h, so a suffix that is a proper
// prefix of another sorts first, exactly like ReadOnlySpan<byte>.SequenceCompareTo.
// ---------------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void gather_key2_kernel(uint64_t *__restrict__ comp,
                                                             const IdxT *__restrict__ suf,
                                                             const IdxT *__restrict__ ISA,
                                                             int64_t m, int64_t n, int64_t h, int kbits,
                                                             int rshift = 0,
                                                             const uint32_t *__restrict__ RL = nullptr /* dq_runs.h, or none */,
                                                             const uint8_t *__restrict__ text = nullptr, int run_order = 0)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < m;
         j += (int64_t)gridDim.x * kBlock) {
        const int64_t s = (int64_t)suf[j];
        int64_t off = h;
        uint64_t k2 = 0;
        bool keyed = false;
        if (RL) {
            // a suffix that starts with a run of >= h equal bytes: the run's own order (run-order round), or the rank
            // behind the run (every later round); the text ending behind the run sorts first
            const uint32_t r = RL[s];
            if (run_order) { k2 = (int64_t)r >= h ? (uint64_t)run_order_key(text, n, s, r, run_order) : 0ull; keyed = true; }
            else if ((int64_t)r > h) off = (int64_t)r;
        }
        if (!keyed) {
            const int64_t q = s + off;
            k2 = q < n ? (uint64_t)((int64_t)ISA[q] + h) : (off > h ? 0ull : (uint64_t)(n - 1 - s));
        }
        comp[j] = ((comp[j] >> rshift) << kbits) | k2;        // rshift = 1: see seg_fused_kernel's rank_from_isa
    }
}

// A list that came keyed for a round (rank << kbits | key2) back to plain group ranks: the round was found to
// need the shifted-rank form of its keys (64-bit composite near n = 2^32, or DQ_FORCE_RSHIFT in the tests).
static __global__ __launch_bounds__(kBlock) void keys_to_ranks_kernel(uint64_t *__restrict__ comp, int64_t m, int kbits)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < m; j += (int64_t)gridDim.x * kBlock)
        comp[j] >>= kbits;
}

// ---------------------------------------------------------------------------------
// Sparse finishing (few suffixes still tied, e.g. random-like inputs): instead of building
// the full inverse suffix array, extend the tied suffixes' keys with the next `ebytes` bytes
// of text:  composite = (rank << kbits) | (bytes, zero padded) << 3 | valid_len,
// kbits = 8*ebytes + 3.  (padded bytes, valid length) orders a suffix that ends inside the
// window before one that continues with real zero bytes -- SequenceCompareTo again.
// ---------------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void gather_text_key_kernel(uint64_t *__restrict__ comp,
                                                                 const IdxT *__restrict__ suf,
                                                                 const uint8_t *__restrict__ text,
                                                                 int64_t m, int64_t n, int64_t h, int ebytes)
{
    const int kbits = 8 * ebytes + 3;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < m;
         j += (int64_t)gridDim.x * kBlock) {
        const int64_t q = (int64_t)suf[j] + h;
        int64_t len = n - q;
        len = len < 0 ? 0 : (len > ebytes ? ebytes : len);
        uint64_t bytes = 0;
        for (int b = 0; b < ebytes; ++b) bytes = (bytes << 8) | (b < len ? (uint64_t)text[q + b] : 0ull);
        comp[j] = (comp[j] << kbits) | (bytes << 3) | (uint64_t)len;
    }
}

// ---------------------------------------------------------------------------------
// small_group_finish: most ties left by round 0 on random-like data are groups of 2-3
// suffixes that differ a few bytes further on.  The first member's lane sorts a group of
// <= kMaxG suffixes by direct text comparison from offset h (at most kMaxLen bytes; a suffix
// that ends first sorts first) and writes SA[rank + i].  What it cannot finish -- groups that
// are larger, or still undecided after kMaxLen bytes -- is appended to the list (out_rank,
// out_suf) for the key-extension rounds: one atomic per workgroup that has leftovers, order
// arbitrary (the next step radix-sorts the list anyway).  *out_count must be zero on entry.
// With m_dev the list length is read on the device (launch without a host round trip).
// ---------------------------------------------------------------------------------
constexpr int kFinishThreads = 256;

template <typename IdxT, int kMaxG, int kMaxLen>
__global__ __launch_bounds__(kFinishThreads) void small_group_finish_kernel(
    const uint64_t *__restrict__ rank, const IdxT *__restrict__ suf, const uint8_t *__restrict__ text,
    int64_t m, int64_t n, int64_t h, IdxT *__restrict__ SA, uint64_t *__restrict__ out_rank,
    IdxT *__restrict__ out_suf, unsigned long long *__restrict__ out_count,
    const unsigned long long *__restrict__ m_dev = nullptr)
{
    // speculative launch (the list length is still on the device): m is the capacity the grid was
    // sized for; a longer list makes the host redo the step, so nothing is done for it here
    if (m_dev) {
        const unsigned long long real = m_dev[0];
        if (real > (unsigned long long)m || m_dev[1] != 0) return;      // m_dev[1]: the producer gave up (TieCounters)
        m = (int64_t)real;
    }
    __shared__ uint32_t wave_tot[kFinishThreads / kWave];
    __shared__ unsigned long long s_base;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    // grid-stride over tiles of kFinishThreads entries (the trip count is uniform per workgroup)
    for (int64_t j0 = (int64_t)blockIdx.x * kFinishThreads; j0 < m; j0 += (int64_t)gridDim.x * kFinishThreads) {
    const int64_t j = j0 + threadIdx.x;

    uint32_t emit = 0;                 // entries this lane appends: itself (large group) or its whole undecided group
    uint64_t r = 0;
    int g = 1;
    bool head = false;
    if (j < m) {
        r = rank[j];
        int left = 0, right = 0;
#pragma unroll
        for (int i = 1; i <= kMaxG; ++i)
            if (left == i - 1 && j - i >= 0 && rank[j - i] == r) left = i;
#pragma unroll
        for (int i = 1; i <= kMaxG; ++i)
            if (right == i - 1 && j + i < m && rank[j + i] == r) right = i;
        g = left + right + 1;
        if (g > kMaxG) emit = 1;
        else head = left == 0;
    }
    int64_t s[kMaxG];
    if (head) {
#pragma unroll
        for (int i = 0; i < kMaxG; ++i) s[i] = i < g ? (int64_t)suf[j + i] : 0;
        // -1: a < b, +1: a > b, 0: undecided within kMaxLen bytes.  The kMaxLen bytes of both suffixes are fetched
        // as 8-byte words, all in flight together (the text is followed by 64 zero bytes, so the words of a suffix
        // that ends inside the window exist; what lies behind an end never decides: `lim` stops the comparison there).
        // (Byte-at-a-time, a pair that shares the whole window cost 2 x kMaxLen dependent loads: a 256 MiB slice of a
        // shared library with 38 M deep ties spent 12.7 ms here.)
        static_assert(kMaxLen % 8 == 0, "whole words");
        typedef uint64_t u64_any __attribute__((aligned(1)));
        auto cmp = [&](int64_t a, int64_t b) -> int {
            const int64_t pa = a + h, pb = b + h;
            uint64_t wa[kMaxLen / 8], wbv[kMaxLen / 8];
#pragma unroll
            for (int k = 0; k < kMaxLen / 8; ++k) {
                wa[k] = *reinterpret_cast<const u64_any *>(text + pa + 8 * k);
                wbv[k] = *reinterpret_cast<const u64_any *>(text + pb + 8 * k);
            }
            int d = kMaxLen;                                   // first differing byte
            bool a_less = false;
#pragma unroll
            for (int k = kMaxLen / 8 - 1; k >= 0; --k) {
                const uint64_t x = __builtin_bswap64(wa[k]), y = __builtin_bswap64(wbv[k]);
                if (x != y) { d = 8 * k + (__builtin_clzll(x ^ y) >> 3); a_less = x < y; }
            }
            const int64_t la = n - pa, lb = n - pb;            // bytes each suffix still has (>= 0)
            const int64_t lim = la < lb ? (la < kMaxLen ? la : kMaxLen) : (lb < kMaxLen ? lb : kMaxLen);
            if (d < lim) return a_less ? -1 : 1;
            if (lim < kMaxLen) return la < lb ? -1 : 1;        // one of them ends inside the window: the shorter first
            return 0;
        };
        // bubble passes with static indices keep s[] in registers.  Nearly every group on random-like
        // data is a pair: it gets its own single comparison instead of walking the 28-site network.
        bool decided = true;
        if (g == 2) {
            const int c = cmp(s[0], s[1]);
            if (c == 0) decided = false;
            if (c > 0) { const int64_t t = s[0]; s[0] = s[1]; s[1] = t; }
        } else
#pragma unroll
        for (int pass = 0; pass < kMaxG - 1; ++pass) {
#pragma unroll
            for (int i = 0; i < kMaxG - 1; ++i) {
                if (i + 1 < g && i < g - 1 - pass) {
                    const int c = cmp(s[i], s[i + 1]);
                    if (c == 0) decided = false;
                    if (c > 0) { const int64_t t = s[i]; s[i] = s[i + 1]; s[i + 1] = t; }
                }
            }
        }
        if (decided) {
#pragma unroll
            for (int i = 0; i < kMaxG; ++i)
                if (i < g) SA[(int64_t)r + i] = (IdxT)s[i];
        } else {
            emit = (uint32_t)g;
        }
    }
    const uint32_t incl = wave_incl_sum(emit);
    if (lane == kWave - 1) wave_tot[w] = incl;
    __syncthreads();
    uint32_t off = incl - emit, tot = 0;
#pragma unroll
    for (int k = 0; k < kFinishThreads / kWave; ++k) {
        const uint32_t c = wave_tot[k];
        if (k < w) off += c;
        tot += c;
    }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(out_count, (unsigned long long)tot) : 0ull;
    __syncthreads();
    if (emit != 0) {
        const int64_t o = (int64_t)s_base + off;
        if (head) {
            // undecided small group: its members as they stand in the list
            for (int i = 0; i < g; ++i) { out_rank[o + i] = r; out_suf[o + i] = suf[j + i]; }
        } else {
            out_rank[o] = r;
            out_suf[o] = suf[j];
        }
    }
    __syncthreads();                   // wave_tot / s_base are reused by the next tile
    }
}

// Estimate of the tie fraction after round 0: kSamples evenly spaced adjacent pairs of the
// sorted key list; *count = pairs with equal keys.  Decides whether the rebucket pass should
// write the full inverse suffix array right away (dense doubling expected).
static __global__ __launch_bounds__(kBlock) void sample_ties_kernel(const uint64_t *__restrict__ keys, int64_t m,
                                                             int kshift, int samples, int64_t *__restrict__ count)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    bool tie = false;
    if (i < samples && m > 1) {
        const int64_t p = (int64_t)((__int128)i * (m - 1) / samples);
        tie = (keys[p] >> kshift) == (keys[p + 1] >> kshift);
    }
    const uint64_t b = __ballot(tie);
    if (lane_id() == 0 && b) atomicAdd((unsigned long long *)count, (unsigned long long)__popcll(b));
}

// ISA[SA[p]] = p : ranks of all singleton groups (switching from sparse to dense doubling)
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void isa_from_sa_kernel(const IdxT *__restrict__ SA,
                                                             IdxT *__restrict__ ISA, int64_t n)
{
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock)
        ISA[SA[p]] = (IdxT)p;
}

// words (p << ib | SA[p]) for the suffix-binned build of the inverse suffix array (dq_isa_pairs.h) from a complete SA
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void sa_words_kernel(const IdxT *__restrict__ SA, int64_t n, int ib,
                                                          uint64_t *__restrict__ words)
{
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock)
        words[p] = ((uint64_t)p << ib) | (uint64_t)SA[p];
}

// ISA[suf[j]] = rank[j] for the still-tied suffixes
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void isa_scatter_kernel(const uint64_t *__restrict__ rank,
                                                             const IdxT *__restrict__ suf,
                                                             IdxT *__restrict__ ISA, int64_t m)
{
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < m; j += (int64_t)gridDim.x * kBlock)
        ISA[suf[j]] = (IdxT)rank[j];
}

}  // namespace dq
