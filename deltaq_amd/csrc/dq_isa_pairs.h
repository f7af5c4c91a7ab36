// dq_isa_pairs.h -- the first inverse suffix array of a dense (text-like) input, without 4-byte
// random scatters over the whole array.
//
// After round 0 every suffix s gets its first rank: ISA[SA[p]] = rank(p) for all n positions, and
// the first doubling round then gathers ISA[s+h] for ~all suffixes again.  Both touch the 4n-byte
// array at random (one 64-byte sector per 4 bytes; n = 2^28: 9.3 ms for the scatter + 5.3 ms for the
// gather on MI355X).  Instead the rebucket pass emits one word per list entry
//      tied? << 63 | rank << ib | suffix               (ib = bits of n-1, needs 2*ib <= 63)
// and two ordinary word passes of the radix sorter (radix_rank_kernel<kKeys>, digit = bits
// [ib-16, ib-8) then [ib-8, ib) of the suffix; the digit histograms are known in closed form because
// every suffix 0..n-1 occurs exactly once) bring the words into "coarse text order": sorted by the
// top 16 bits of the suffix.  Then
//   isa_from_pairs_kernel      ISA[suffix] = rank: the words of one span of 4096 (or 2^(ib-16)) consecutive
//                              suffixes are consecutive too, so a workgroup assembles that piece of the ISA
//                              in LDS and writes it with full coalesced lines
//   key2_from_pairs_kernel     for the tied suffixes: key2 = ISA[s+h] + h | n-1-s, read inside the same
//                              kind of window, appended as (rank << kbits | key2, s) for the first
//                              doubling round (which then skips its gather), or as (rank, s) when the
//                              round will use the small-group kernel; one atomic per workgroup, order
//                              arbitrary (the round sorts the list)
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kPairThreads = 1024;
constexpr int kPairItems = 4;

// Words sorted by the top 16 bits of the suffix, every suffix 0..n-1 present exactly once: the words at
// positions [v*S, (v+1)*S) are exactly the suffixes v*S .. (v+1)*S-1 in some order (S = span, a multiple
// of the bin width 2^(ib-16)).  One workgroup takes one span: ranks go to their place in an LDS image of
// ISA[v*S ..) and the image is written out with full, coalesced lines.
template <typename IdxT, int kSpan>
__global__ __launch_bounds__(kPairThreads) void isa_from_pairs_kernel(const uint64_t *__restrict__ pairs, int64_t n,
                                                                    int ib, IdxT *__restrict__ ISA)
{
    __shared__ uint32_t image[kSpan];                 // ranks < 2^ib <= 2^31
    const uint64_t mask = (1ull << ib) - 1;
    const int64_t base = (int64_t)blockIdx.x * kSpan;
    for (int i = threadIdx.x; i < kSpan; i += kPairThreads) {
        if (base + i < n) {
            const uint64_t w = pairs[base + i];
            image[(int64_t)(w & mask) - base] = (uint32_t)((w >> ib) & mask);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSpan; i += kPairThreads)
        if (base + i < n) ISA[base + i] = (IdxT)image[i];
}

// Rank updates of a doubling round, applied window by window (round 5).  `words` = (rank << ib | suffix), sorted by the
// top 16 bits of the suffix (two word passes of the radix sorter), so the updates of one span of kSpan consecutive
// suffixes are consecutive; bounds[v] = index of the first word of span v (window_bounds_kernel).  A workgroup loads its
// span of the inverse suffix array into LDS, applies the span's updates there and writes the span back in full lines --
// 8 bytes of array traffic per suffix of a touched span instead of one read-modify-write of a 64-byte sector per update
// (isa_update_words_kernel on words binned by 8 bits: 8x write amplification, rocprof WRITE_SIZE, round 4).  Pays while
// at least ~1/16 of the array is updated.  The all-ones word is the alignment filler in front of the list and is skipped
// (a real word is rank << ib | suffix with 2 * ib <= 63: bit 63 clear.  Its suffix field alone does not tell: for n = 2^ib
// it reads n - 1, a real suffix).
template <typename IdxT, int kSpan>
__global__ __launch_bounds__(kPairThreads) void isa_update_window_kernel(const uint64_t *__restrict__ words, const int64_t *__restrict__ bounds,
                                                                       int64_t n, int ib, IdxT *__restrict__ ISA)
{
    __shared__ uint32_t image[kSpan];
    const int64_t lo = bounds[blockIdx.x], hi = bounds[blockIdx.x + 1];
    if (lo >= hi) return;                                 // nothing moved in this span
    const uint64_t mask = (1ull << ib) - 1;
    const int64_t base = (int64_t)blockIdx.x * kSpan;
    for (int i = threadIdx.x; i < kSpan; i += kPairThreads)
        if (base + i < n) image[i] = (uint32_t)ISA[base + i];
    __syncthreads();
    for (int64_t i = lo + threadIdx.x; i < hi; i += kPairThreads) {
        const uint64_t w = words[i];
        const int64_t sfx = (int64_t)(w & mask);
        if (w != ~0ull && sfx < n) image[sfx - base] = (uint32_t)(w >> ib);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSpan; i += kPairThreads)
        if (base + i < n) ISA[base + i] = (IdxT)image[i];
}

// bounds[v] = first index whose word lies in span >= v (v = 0 .. nspans), by binary search over the sorted words
static __global__ __launch_bounds__(kBlock) void window_bounds_kernel(const uint64_t *__restrict__ words, int64_t count, int ib,
                                                               int span_log2, int64_t nspans, int64_t *__restrict__ bounds)
{
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v > nspans) return;
    const uint64_t mask = (1ull << ib) - 1;
    int64_t lo = 0, hi = count;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)((words[mid] & mask) >> span_log2) < v) lo = mid + 1;
        else hi = mid;
    }
    bounds[v] = lo;
}

// *count must be zero on entry; it ends as the number of tied suffixes appended
template <typename IdxT>
__global__ __launch_bounds__(kPairThreads) void key2_from_pairs_kernel(
    const uint64_t *__restrict__ pairs, int64_t n, int ib, const IdxT *__restrict__ ISA, int64_t h, int kbits,
    bool with_key2, uint64_t *__restrict__ out_key, IdxT *__restrict__ out_suf, unsigned long long *__restrict__ count)
{
    __shared__ uint32_t wave_cnt[kPairItems][kPairThreads / kWave];
    __shared__ unsigned long long s_base;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const uint64_t mask = (1ull << ib) - 1;
    const int64_t base = (int64_t)blockIdx.x * (kPairThreads * kPairItems);
    uint64_t word[kPairItems], key[kPairItems], bal[kPairItems];
#pragma unroll
    for (int k = 0; k < kPairItems; ++k) {
        const int64_t i = base + k * kPairThreads + threadIdx.x;
        word[k] = i < n ? pairs[i] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < kPairItems; ++k) {
        const bool tied = (word[k] >> 63) != 0;
        key[k] = 0;
        if (tied) {
            const int64_t s = (int64_t)(word[k] & mask);
            const uint64_t r = (word[k] >> ib) & mask;
            if (with_key2) {
                const int64_t q = s + h;
                const uint64_t k2 = q < n ? (uint64_t)((int64_t)ISA[q] + h) : (uint64_t)(n - 1 - s);   // as gather_key2_kernel
                key[k] = (r << kbits) | k2;
            } else {
                key[k] = r;
            }
        }
        bal[k] = __ballot(tied);
        if (lane == 0) wave_cnt[k][w] = (uint32_t)__popcll(bal[k]);
    }
    __syncthreads();
    if (w == 0) {
        // exclusive scan of the kPairItems x 16 wave counts in (item, wave) order
        uint32_t *c = &wave_cnt[0][0];
        const uint32_t v = lane < kPairItems * (kPairThreads / kWave) ? c[lane] : 0u;
        const uint32_t incl = wave_incl_sum(v);
        if (lane < kPairItems * (kPairThreads / kWave)) c[lane] = incl - v;
        const uint32_t tot = __shfl(incl, kWave - 1, kWave);
        if (lane == 0) s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPairItems; ++k) {
        if (word[k] >> 63) {
            const int64_t o = (int64_t)s_base + wave_cnt[k][w] + mask_rank_lt(bal[k]);
            out_key[o] = key[k];
            out_suf[o] = (IdxT)(word[k] & mask);
        }
    }
}

// ranks of the tied list, written as 32-bit values while every 64-bit buffer was busy, into their 64-bit slots
static __global__ __launch_bounds__(kBlock) void widen_ranks_kernel(const uint32_t *__restrict__ in, int64_t m,
                                                             uint64_t *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock)
        out[i] = (uint64_t)in[i];
}

}  // namespace dq
