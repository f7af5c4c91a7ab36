// dq_small.h -- the whole suffix sort of a short text (n <= kSmallMaxN) in ONE workgroup.
//
// The reference's own benchmark and fixtures live here (SuffixSortingBenchmarks.cs:27-53 sizes
// 64 B .. 32 KiB; test/assets/* are 17 B .. 4.8 kB with LCPs in the thousands).  The device-wide
// pipeline costs ~10 launches and several host round trips per doubling round, which is all
// overhead at this size, so short texts get a single launch: prefix doubling with every array in
// LDS, a workgroup-wide stable LSD radix sort per round (per-wave ballot ranking + per-wave digit
// counters, the same scheme as radix_rank_kernel minus the inter-workgroup protocol), and no host
// interaction until the SA is complete.
//
// Same ordering rules as the large path: rank = SA index of the group's first
// member; key2 = ISA[s+h] + h if s+h < n else n-1-s, so a proper prefix sorts first
// (ReadOnlySpan<byte>.SequenceCompareTo, LibDivSufSortTests.cs:43-59).
#pragma once
#include "dq_device_utils.h"
#include "dq_runtime.h"

namespace dq {

// (kSmallMaxN = 8192 lives in dq_runtime.h: the host runtime sizes its pinned areas by it)
constexpr int kSmallThreads = 1024;
constexpr int kSmallWaves = kSmallThreads / kWave;
constexpr int kSmallItems = kSmallMaxN / kSmallThreads;       // positions per thread at the largest n

struct SmallLds {
    uint32_t key[2][kSmallMaxN];          // composite keys, ping-pong                       64 KiB
    uint16_t val[2][kSmallMaxN];          // suffix indices, ping-pong                       32 KiB
    uint16_t isa[kSmallMaxN + 8];         // ranks by text position (holds the text first)   16 KiB
    uint16_t cnt[kSmallWaves][256];       // per-wave digit counts -> scatter bases           8 KiB
    int32_t wmax[kSmallWaves];
    int32_t wsum[kSmallWaves];
    uint32_t dsum[4];
};

// One stable 8-bit digit pass src -> dst over positions [0, n).  Wave w owns the contiguous
// positions [w*64*E, (w+1)*64*E) and walks them 64 at a time, so "earlier position" is
// (earlier wave, earlier step, lower lane).
__device__ __forceinline__ void small_digit_pass(SmallLds &L, int src, int n, int E, int shift)
{
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int dst = src ^ 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) L.cnt[w][lane + 64 * i] = 0;
    // (same-wave LDS operations complete in program order: no barrier needed before the counting)
    uint32_t kreg[kSmallItems];
    uint16_t vreg[kSmallItems];
    uint16_t local[kSmallItems];
    const int base = w * 64 * E;
#pragma unroll
    for (int k = 0; k < kSmallItems; ++k) {
        if (k < E) {
            const int p = base + k * 64 + lane;
            const bool valid = p < n;
            kreg[k] = valid ? L.key[src][p] : 0xffffffffu;
            vreg[k] = valid ? L.val[src][p] : 0;
            const uint32_t d = (kreg[k] >> shift) & 255u;
            const uint64_t same = match_digit8(d) & __ballot(valid);
            const int before = mask_rank_lt(same);
            const uint16_t prev = L.cnt[w][d];
            local[k] = (uint16_t)(prev + before);
            if (valid && before == 0) L.cnt[w][d] = (uint16_t)(prev + __popcll(same));
        }
    }
    __syncthreads();
    // exclusive scan of cnt in (digit, wave) order: thread d < 256 walks the 16 waves of digit d
    uint32_t tot = 0;
    uint16_t c[kSmallWaves];
    if (threadIdx.x < 256) {
#pragma unroll
        for (int i = 0; i < kSmallWaves; ++i) c[i] = L.cnt[i][threadIdx.x];
#pragma unroll
        for (int i = 0; i < kSmallWaves; ++i) { const uint16_t t = c[i]; c[i] = (uint16_t)tot; tot += t; }
        const uint32_t incl = wave_incl_sum(tot);
        if (lane == 63) L.dsum[w] = incl;
        tot = incl - tot;                                   // exclusive inside this wave of digits
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        for (int i = 0; i < w; ++i) tot += L.dsum[i];
#pragma unroll
        for (int i = 0; i < kSmallWaves; ++i) L.cnt[i][threadIdx.x] = (uint16_t)(c[i] + tot);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSmallItems; ++k) {
        if (k < E) {
            const int p = base + k * 64 + lane;
            if (p < n) {
                const uint32_t d = (kreg[k] >> shift) & 255u;
                const int pos = L.cnt[w][d] + local[k];
                L.key[dst][pos] = kreg[k];
                L.val[dst][pos] = vreg[k];
            }
        }
    }
    __syncthreads();
}

// Group heads of the sorted list in buffer `cur`, rank = position of the group's head,
// isa[suffix] = rank.  Returns the number of groups (uniform over the workgroup).
__device__ __forceinline__ int small_rebucket(SmallLds &L, int cur, int n, int E)
{
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int first = threadIdx.x * E;                          // blocked ownership for the scan
    int lastHead[kSmallItems];
    int m = -1, heads = 0;
#pragma unroll
    for (int k = 0; k < kSmallItems; ++k) {
        if (k < E) {
            const int p = first + k;
            if (p < n && (p == 0 || L.key[cur][p] != L.key[cur][p - 1])) { m = p; ++heads; }
            lastHead[k] = m;
        }
    }
    const int incl = wave_incl_max(m);
    int excl = __shfl_up(incl, 1, kWave);
    if (lane == 0) excl = -1;
    const int hs = wave_sum(heads);
    if (lane == 63) L.wmax[w] = incl;
    if (lane == 0) L.wsum[w] = hs;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int i = 0; i < kSmallWaves; ++i) {
        if (i < w) excl = max(excl, L.wmax[i]);
        total += L.wsum[i];
    }
#pragma unroll
    for (int k = 0; k < kSmallItems; ++k) {
        if (k < E) {
            const int p = first + k;
            if (p < n) L.isa[L.val[cur][p]] = (uint16_t)(lastHead[k] >= 0 ? lastHead[k] : excl);
        }
    }
    __syncthreads();
    return total;
}

__device__ __forceinline__ int small_bits(uint32_t x) { return x ? 32 - __builtin_clz(x) : 0; }

template <typename IdxT>
__global__ __launch_bounds__(kSmallThreads) void small_sufsort_kernel(const uint8_t *__restrict__ text, int n,
                                                                      IdxT *__restrict__ sa)
{
    __shared__ SmallLds L;
    const int t = threadIdx.x;
    const int E = (n + kSmallThreads - 1) / kSmallThreads;

    // the text, zero padded, parked in the (not yet used) isa array
    uint8_t *T = reinterpret_cast<uint8_t *>(L.isa);
    for (int i = t; i < n + 4; i += kSmallThreads) T[i] = i < n ? text[i] : (uint8_t)0;
    __syncthreads();
    for (int i = t; i < n; i += kSmallThreads) {
        L.key[0][i] = ((uint32_t)T[i] << 24) | ((uint32_t)T[i + 1] << 16) | ((uint32_t)T[i + 2] << 8) | T[i + 3];
        L.val[0][i] = (uint16_t)i;
    }
    __syncthreads();
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) { small_digit_pass(L, cur, n, E, shift); cur ^= 1; }
    int groups = small_rebucket(L, cur, n, E);

    const int rbits = small_bits((uint32_t)(n - 1));
    for (int h = 4; groups < n; h *= 2) {
        // ties need s+h < n for both suffixes, so h < n here and key2 < 2n
        const int kbits = small_bits((uint32_t)(n - 1 + h));
        for (int p = t; p < n; p += kSmallThreads) {
            const int s = L.val[cur][p];
            const int q = s + h;
            const uint32_t k2 = q < n ? (uint32_t)L.isa[q] + (uint32_t)h : (uint32_t)(n - 1 - s);
            L.key[cur][p] = ((uint32_t)L.isa[s] << kbits) | k2;
        }
        __syncthreads();
        for (int shift = 0; shift < rbits + kbits; shift += 8) { small_digit_pass(L, cur, n, E, shift); cur ^= 1; }
        groups = small_rebucket(L, cur, n, E);
    }
    for (int p = t; p < n; p += kSmallThreads) sa[p] = (IdxT)L.val[cur][p];
}

}  // namespace dq
