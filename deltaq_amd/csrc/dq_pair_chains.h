// dq_pair_chains.h -- tie groups of exactly two suffixes inside long repeats, finished in ONE phase.
//
// A repeat of L bytes at text positions p and q leaves L tied pairs {p+k, q+k}; prefix doubling needs
// log2(L / h) more rounds for them, each a pass of random rank gathers over all of them (enwik-like 256 MiB:
// 3e7 suffixes for 11 rounds).  But once two suffixes a < b agree on their first character (h >= 1),
//      order(a, b) = order(a+1, b+1),
// and {a+1, b+1} is either decided already (different ranks), or the next pair of the same chain.  So a chain
// has ONE answer, found at its last pair, and every pair of the chain copies it:
//
//   pair_split_kernel<false/true>  over the tied list X (members of a group adjacent): groups of exactly 2 become
//                       records (b << 32 | a, rank) with a < b; everything else is copied, in order, to the
//                       list of the following doubling rounds (count pass + scan + write pass: stable)
//   (radix sort)        records by a (onesweep_sort_pairs on the low bits of the word)
//   pair_link_kernel    record i is a LINK if record i+1 is (a+1, b+1); otherwise it is the END of a chain and
//                       decides it: ISA[a+1] < ISA[b+1] (a first), > (b first).  Equal ranks: a+1 and b+1 sit in
//                       one tie group of >= h characters (boilerplate inside the repeat, shared with other
//                       places): if that group is itself a pair, this chain takes ITS answer (a FAR link to a
//                       record further right); a larger group is stepped over h characters at a time; failing
//                       that the chain is BLOCKED and goes back to doubling.  Every record learns the index of
//                       the first chain end at or after it (segmented scan from the right, tile-local here)
//   pair_carry_kernel   ... and across tiles
//   pair_resolve_kernel x6: pointer jumping over the far links (they only lead to the right)
//   pair_emit_kernel    decided pairs: SA[rank], SA[rank+1] and the ISA entry of the second; blocked pairs:
//                       appended to the list behind the copied entries
//
// Reads of the ISA (link kernel) and writes to it (emit kernel) are separate launches.  Records keep 32-bit
// suffix halves: n <= 2^32 (check_args).  Nothing here depends on h beyond h >= 1.
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kPcThreads = 256;
constexpr int kPcItems = 8;
constexpr int kPcTile = kPcThreads * kPcItems;
constexpr int kPcScanThreads = 1024;

struct PairCounters {
    unsigned long long pairs;      // records made
    unsigned long long list;       // entries of the next list: copied ones, then the blocked pairs appended
};

// One pass over X in tiles of kPcTile entries.  kWrite = false: tile_cnt[2t] = pairs, [2t+1] = other entries of
// tile t.  kWrite = true: tile_cnt holds the exclusive prefix sums and the records / copies are written.
template <typename IdxT, bool kWrite>
__global__ __launch_bounds__(kPcThreads) void pair_split_kernel(const uint64_t *__restrict__ rank,
                                                                const IdxT *__restrict__ suf, int64_t m,
                                                                uint32_t *__restrict__ tile_cnt,
                                                                uint64_t *__restrict__ rec_key, IdxT *__restrict__ rec_rank,
                                                                uint64_t *__restrict__ out_rank, IdxT *__restrict__ out_suf)
{
    __shared__ uint64_t s_rank[kPcTile + 4];
    __shared__ uint32_t wave_cnt[2][kPcItems][kPcThreads / kWave];
    const int t = threadIdx.x, lane = lane_id(), wv = t >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * kPcTile;
    constexpr uint64_t kNone = ~0ull;
    for (int e = t; e < kPcTile + 4; e += kPcThreads) {
        const int64_t j = j0 - 2 + e;
        s_rank[e] = (j >= 0 && j < m) ? rank[j] : kNone;
    }
    __syncthreads();
    bool first[kPcItems], other[kPcItems];
    uint64_t bf[kPcItems], bo[kPcItems];
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        const uint64_t r = s_rank[e + 2], l1 = s_rank[e + 1], l2 = s_rank[e], r1 = s_rank[e + 3], r2 = s_rank[e + 4];
        const bool valid = j0 + e < m;
        first[k] = valid && l1 != r && r1 == r && r2 != r;
        const bool second = valid && l1 == r && l2 != r && r1 != r;
        other[k] = valid && !first[k] && !second;
        bf[k] = __ballot(first[k]);
        bo[k] = __ballot(other[k]);
        if (lane == 0) {
            wave_cnt[0][k][wv] = (uint32_t)__popcll(bf[k]);
            wave_cnt[1][k][wv] = (uint32_t)__popcll(bo[k]);
        }
    }
    __syncthreads();
    // exclusive scan of the 2 x 32 (item, wave) counts in position order: wave 0 takes the pairs, wave 1 the others
    __shared__ uint32_t tot[2];
    if (wv < 2) {
        uint32_t *c = &wave_cnt[wv][0][0];
        const uint32_t v = lane < kPcItems * (kPcThreads / kWave) ? c[lane] : 0u;
        const uint32_t incl = wave_incl_sum(v);
        if (lane < kPcItems * (kPcThreads / kWave)) c[lane] = incl - v;
        if (lane == kWave - 1) tot[wv] = incl;
    }
    __syncthreads();
    if (!kWrite) {
        if (t < 2) tile_cnt[2 * (int64_t)blockIdx.x + t] = tot[t];
        return;
    }
    const int64_t base_p = tile_cnt[2 * (int64_t)blockIdx.x], base_o = tile_cnt[2 * (int64_t)blockIdx.x + 1];
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        const int64_t j = j0 + e;
        if (first[k]) {
            const uint64_t x = (uint64_t)suf[j], y = (uint64_t)suf[j + 1];
            const uint64_t a = x < y ? x : y, b = x < y ? y : x;
            const int64_t o = base_p + wave_cnt[0][k][wv] + mask_rank_lt(bf[k]);
            rec_key[o] = (b << 32) | a;
            rec_rank[o] = (IdxT)s_rank[e + 2];
        }
        if (other[k]) {
            const int64_t o = base_o + wave_cnt[1][k][wv] + mask_rank_lt(bo[k]);
            out_rank[o] = s_rank[e + 2];
            out_suf[o] = suf[j];
        }
    }
}

// tile_cnt[2t], tile_cnt[2t+1] -> exclusive prefix sums over the tiles (two independent sums), totals to ctr.
// One workgroup: every thread sums a contiguous run of tiles, the runs are scanned, the run is rewritten.
__global__ __launch_bounds__(kPcScanThreads) void pair_scan_kernel(uint32_t *__restrict__ tile_cnt, int64_t ntiles,
                                                                   PairCounters *__restrict__ ctr)
{
    __shared__ unsigned long long part[2][kPcScanThreads / kWave];
    const int t = threadIdx.x, lane = lane_id(), wv = t >> 6;
    const int64_t per = (ntiles + kPcScanThreads - 1) / kPcScanThreads;
    const int64_t lo = (int64_t)t * per, hi = lo + per < ntiles ? lo + per : ntiles;
    unsigned long long sp = 0, so = 0;
    for (int64_t i = lo; i < hi; ++i) { sp += tile_cnt[2 * i]; so += tile_cnt[2 * i + 1]; }
    const unsigned long long ip = wave_incl_sum(sp), io = wave_incl_sum(so);
    if (lane == kWave - 1) { part[0][wv] = ip; part[1][wv] = io; }
    __syncthreads();
    unsigned long long bp = ip - sp, bo = io - so, tp = 0, to = 0;
#pragma unroll
    for (int i = 0; i < kPcScanThreads / kWave; ++i) {
        if (i < wv) { bp += part[0][i]; bo += part[1][i]; }
        tp += part[0][i];
        to += part[1][i];
    }
    for (int64_t i = lo; i < hi; ++i) {
        const uint32_t cp = tile_cnt[2 * i], co = tile_cnt[2 * i + 1];
        tile_cnt[2 * i] = (uint32_t)bp;
        tile_cnt[2 * i + 1] = (uint32_t)bo;
        bp += cp;
        bo += co;
    }
    if (t == 0) { ctr->pairs = tp; ctr->list = to; }
}

// status of a chain end: 1 a first, 2 b first, 3 blocked, 4 far (takes the answer of record far[i], further right)
constexpr uint8_t kPcAFirst = 1, kPcBFirst = 2, kPcBlocked = 3, kPcFar = 4;
constexpr uint32_t kPcNone = 0xffffffffu;
constexpr int kPcJumps = 4;               // h-steps a chain end tries through larger tie groups
constexpr int kPcResolveRounds = 6;       // pointer-jumping launches: far links nested up to 2^6 deep

// In a wave: the first value != kPcNone among the lanes ABOVE the calling lane (kPcNone if there is none).
__device__ __forceinline__ uint32_t first_valid_above(uint32_t v)
{
    const uint64_t nz = __ballot(v != kPcNone);
    const int lane = lane_id();
    const uint64_t above = lane == kWave - 1 ? 0ull : (nz >> (lane + 1)) << (lane + 1);
    const int src = above ? __builtin_ctzll(above) : lane;
    const uint32_t got = (uint32_t)__shfl((int)v, src, kWave);
    return above ? got : kPcNone;
}

// Record i is a LINK if record i+1 is (a+1, b+1).  Otherwise it ends a chain and is evaluated:
//   x = a+1, y = b+1 (both suffixes agree on >= 1 character, so order(a, b) = order(x, y));
//   y = n (empty suffix): b first.  ISA[x] != ISA[y]: decided.  Same tie group (>= h equal characters): if the
//   group is the pair {x, y}, its record (found by binary search: the records are sorted by a) answers for this
//   chain too -- a FAR link, always to the right; else step h characters further, a few times; else blocked.
// nt[i] = index of the first chain end at or after i inside the tile (kPcNone: none), tile_head = nt of the
// tile's first record.
template <typename IdxT>
__global__ __launch_bounds__(kPcThreads) void pair_link_kernel(const uint64_t *__restrict__ rec_key, int64_t cnt,
                                                               const IdxT *__restrict__ ISA, int64_t n, int64_t h,
                                                               uint32_t *__restrict__ nt, uint8_t *__restrict__ tstat,
                                                               uint32_t *__restrict__ far, uint32_t *__restrict__ tile_head)
{
    __shared__ uint32_t s_nt[kPcTile];
    __shared__ uint32_t wave_first[kPcThreads / kWave];
    const int t = threadIdx.x, lane = lane_id(), wv = t >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kPcTile;
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        const int64_t i = i0 + e;
        uint32_t mine = kPcNone;
        if (i < cnt) {
            const uint64_t key = rec_key[i];
            const bool link = i + 1 < cnt && rec_key[i + 1] == key + 0x100000001ull;
            if (!link) {
                mine = (uint32_t)i;
                int64_t x = (int64_t)(key & 0xffffffffull) + 1, y = (int64_t)(key >> 32) + 1;
                uint8_t s = kPcBlocked;
                uint32_t target = 0;
                for (int j = 0; j <= kPcJumps; ++j) {
                    if (y >= n) { s = kPcBFirst; break; }             // a < b: y runs out first, the shorter suffix is smaller
                    const int64_t rx = (int64_t)ISA[x], ry = (int64_t)ISA[y];
                    if (rx != ry) { s = rx < ry ? kPcAFirst : kPcBFirst; break; }
                    int64_t lo = i + 1, hi = cnt;                     // lower bound of x among the a's to the right
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if ((int64_t)(rec_key[mid] & 0xffffffffull) < x) lo = mid + 1; else hi = mid;
                    }
                    if (lo < cnt && rec_key[lo] == (((uint64_t)y << 32) | (uint64_t)x)) { s = kPcFar; target = (uint32_t)lo; break; }
                    x += h;
                    y += h;
                }
                tstat[i] = s;
                far[i] = target;
            }
        }
        s_nt[e] = mine;
    }
    __syncthreads();
    // thread t owns positions [8t, 8t+8): right-to-left inside the run, then across the lanes and waves
    uint32_t v[kPcItems];
    uint32_t run = kPcNone;                              // first chain end of the run
#pragma unroll
    for (int k = kPcItems - 1; k >= 0; --k) {
        v[k] = s_nt[t * kPcItems + k];
        if (v[k] != kPcNone) run = v[k];
    }
    uint32_t in = first_valid_above(run);                // what enters the run from the right, inside the wave
    {
        const uint64_t nz = __ballot(run != kPcNone);
        const uint32_t wf = nz ? (uint32_t)__shfl((int)run, __builtin_ctzll(nz), kWave) : kPcNone;
        if (lane == 0) wave_first[wv] = wf;
    }
    __syncthreads();
    if (in == kPcNone) {
#pragma unroll
        for (int i = kPcThreads / kWave - 1; i >= 0; --i)
            if (i > wv && wave_first[i] != kPcNone) in = wave_first[i];
    }
    uint32_t cur = in;
#pragma unroll
    for (int k = kPcItems - 1; k >= 0; --k) {
        if (v[k] != kPcNone) cur = v[k];
        s_nt[t * kPcItems + k] = cur;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        if (i0 + e < cnt) nt[i0 + e] = s_nt[e];
    }
    if (t == 0) tile_head[blockIdx.x] = s_nt[0];
}

// carry[t] = first chain end of the tiles behind tile t (one workgroup, as pair_scan_kernel)
__global__ __launch_bounds__(kPcScanThreads) void pair_carry_kernel(const uint32_t *__restrict__ tile_head, int64_t ntiles,
                                                                    uint32_t *__restrict__ carry)
{
    __shared__ uint32_t wave_first[kPcScanThreads / kWave];
    const int t = threadIdx.x, lane = lane_id(), wv = t >> 6;
    const int64_t per = (ntiles + kPcScanThreads - 1) / kPcScanThreads;
    const int64_t lo = (int64_t)t * per, hi = lo + per < ntiles ? lo + per : ntiles;
    uint32_t run = kPcNone;
    for (int64_t i = hi - 1; i >= lo; --i) { const uint32_t hd = tile_head[i]; if (hd != kPcNone) run = hd; }
    uint32_t in = first_valid_above(run);
    {
        const uint64_t nz = __ballot(run != kPcNone);
        const uint32_t wf = nz ? (uint32_t)__shfl((int)run, __builtin_ctzll(nz), kWave) : kPcNone;
        if (lane == 0) wave_first[wv] = wf;
    }
    __syncthreads();
    if (in == kPcNone) {
        for (int i = kPcScanThreads / kWave - 1; i >= 0; --i)
            if (i > wv && wave_first[i] != kPcNone) in = wave_first[i];
    }
    uint32_t cur = in;
    for (int64_t i = hi - 1; i >= lo; --i) {
        carry[i] = cur;                                   // what enters tile i from the right
        const uint32_t hd = tile_head[i];
        if (hd != kPcNone) cur = hd;
    }
}

__device__ __forceinline__ uint32_t pc_chain_end(const uint32_t *__restrict__ nt, const uint32_t *__restrict__ carry, int64_t i)
{
    const uint32_t e = nt[i];
    return e != kPcNone ? e : carry[i / kPcTile];
}

// One pointer-jumping step for the FAR chain ends: take the answer of the chain the link points into, or, if
// that chain ends FAR as well, point where it points.  (Links only lead to the right, so racing reads see either
// the old or the new link of a neighbour, both valid.)
__global__ __launch_bounds__(kPcThreads) void pair_resolve_kernel(const uint32_t *__restrict__ nt,
                                                                  const uint32_t *__restrict__ carry, int64_t cnt,
                                                                  uint8_t *__restrict__ tstat, uint32_t *__restrict__ far)
{
    const int64_t i = (int64_t)blockIdx.x * kPcThreads + threadIdx.x;
    if (i >= cnt || nt[i] != (uint32_t)i) return;         // chain ends only
    volatile uint8_t *vs = tstat;
    volatile uint32_t *vf = far;
    if (vs[i] != kPcFar) return;
    const uint32_t e = pc_chain_end(nt, carry, (int64_t)vf[i]);
    const uint8_t s = vs[e];
    if (s != kPcFar) vs[i] = s; else vf[i] = vf[e];
}

template <typename IdxT>
__global__ __launch_bounds__(kPcThreads) void pair_emit_kernel(const uint64_t *__restrict__ rec_key,
                                                               const IdxT *__restrict__ rec_rank, int64_t cnt,
                                                               const uint32_t *__restrict__ nt, const uint32_t *__restrict__ carry,
                                                               const uint8_t *__restrict__ tstat, IdxT *__restrict__ SA,
                                                               IdxT *__restrict__ ISA, uint64_t *__restrict__ out_rank,
                                                               IdxT *__restrict__ out_suf, PairCounters *__restrict__ ctr)
{
    const int64_t i = (int64_t)blockIdx.x * kPcThreads + threadIdx.x;
    uint32_t ans = 0;
    uint64_t key = 0;
    int64_t r = 0;
    if (i < cnt) {
        ans = tstat[pc_chain_end(nt, carry, i)];
        if (ans == kPcFar) ans = kPcBlocked;              // nested deeper than the resolve rounds reach: back to doubling
        key = rec_key[i];
        r = (int64_t)rec_rank[i];
    }
    const IdxT a = (IdxT)(key & 0xffffffffull), b = (IdxT)(key >> 32);
    if (ans == kPcAFirst) { SA[r] = a; SA[r + 1] = b; ISA[(int64_t)(key >> 32)] = (IdxT)(r + 1); }
    if (ans == kPcBFirst) { SA[r] = b; SA[r + 1] = a; ISA[(int64_t)(key & 0xffffffffull)] = (IdxT)(r + 1); }
    // blocked pairs: appended behind the copied entries, ONE atomic per workgroup
    __shared__ uint32_t wave_cnt[kPcThreads / kWave];
    __shared__ unsigned long long s_base;
    const uint64_t blocked = __ballot(ans == kPcBlocked);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 0) wave_cnt[wv] = (uint32_t)__popcll(blocked);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < kPcThreads / kWave; ++i) { const uint32_t c = wave_cnt[i]; wave_cnt[i] = tot; tot += c; }
        s_base = tot ? atomicAdd(&ctr->list, 2ull * tot) : 0ull;
    }
    __syncthreads();
    if (ans == kPcBlocked) {
        const int64_t o = (int64_t)s_base + 2 * ((int64_t)wave_cnt[wv] + mask_rank_lt(blocked));
        out_rank[o] = (uint64_t)r; out_suf[o] = a;
        out_rank[o + 1] = (uint64_t)r; out_suf[o + 1] = b;
    }
}

}  // namespace dq
