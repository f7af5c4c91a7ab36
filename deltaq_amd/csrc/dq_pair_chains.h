// dq_pair_chains.h -- small tie groups inside long repeats, finished in ONE phase instead of log2(LCP / h) rounds.
//
// A repeat of L bytes at text positions p and q leaves L tied pairs {p+k, q+k}; prefix doubling needs
// log2(L / h) more rounds for them, each a pass of random rank gathers over all of them (enwik-like 256 MiB:
// 3e7 suffixes for 11 rounds).  But once two suffixes x < y agree on their first character (h >= 1),
//      order(x, y) = order(x+1, y+1),
// and {x+1, y+1} is either decided already (different ranks), or the next pair of the same chain.  So a chain
// of pairs with one distance d = y - x has ONE answer, found at its last pair, and every pair of the chain
// copies it.  A repeat copied more than once leaves groups of 3 or 4; the same holds for each of their 3 or 6
// member pairs, and a group whose pairs are all decided is sorted.
//
//   pair_split_kernel<false/true>  over the tied list X (members of a group adjacent): every group of <= kPcMaxG
//                       members becomes its pairs: records (d << ib | x, ordinal) with x < y = x + d, numbered in
//                       list order; members of larger groups are copied, in order, to the list of the following
//                       doubling rounds (count pass + scan + write pass: stable)
//   (radix sort)        records by (d, x): (x+1, y+1) is then the NEXT record, if it is a record at all
//   pair_link_kernel    record i is a LINK if record i+1 is its key + 1; otherwise it is the END of a chain and
//                       decides it: ISA[x+1] < ISA[y+1] (x first), > (y first).  Equal ranks: x+1 and y+1 sit in
//                       one tie group of >= h characters (boilerplate inside the repeat, shared with other
//                       places): if that group is small, the record of {x+1, y+1} exists further right (binary
//                       search) and this chain takes ITS answer (a FAR link); a large group is stepped over h
//                       characters at a time; failing that the chain is BLOCKED.  Every record learns the index
//                       of the first chain end at or after it (segmented scan from the right, tile-local here)
//   pair_carry_kernel   ... and across tiles
//   pair_resolve_kernel x8: pointer jumping over the far links (they only lead to the right)
//   pair_answer_kernel  answers back into list order (by ordinal)
//   pair_finish_kernel  over X again: a small group whose pairs are all decided writes its members to the SA (place
//                       = number of members that precede it) and the ISA entries that moved; any other small
//                       group is appended, whole, behind the copied entries
//   pair_emit_kernel    pairs-only mode (long lists; records sorted by x alone, four digit passes): answer + finish
//                       in one, in record order
//
// Reads of the ISA (link kernel) and writes to it (finish kernel) are separate launches.  Records need
// 2 * ib <= 64 bits (n <= 2^32: check_args).  Nothing here depends on h beyond h >= 1.
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kPcThreads = 256;
constexpr int kPcItems = 8;
constexpr int kPcTile = kPcThreads * kPcItems;
constexpr int kPcScanThreads = 1024;
constexpr int kPcMaxG = 4;                // groups of up to 4 members (6 pairs); the caller may ask for pairs only
constexpr int kPcMaxPairs = kPcMaxG * (kPcMaxG - 1) / 2;

struct PairCounters {
    unsigned long long pairs;      // records made
    unsigned long long list;       // entries of the next list: copied ones, then the undecided small groups appended
};

// What one list position is, from the ranks around it (s_rank holds the tile with kPcMaxG entries of halo on each
// side, position e of the tile at s_rank[e + kPcMaxG]): size of its group capped at kPcMaxG + 1, and whether it
// is the group's first member.
__device__ __forceinline__ void pc_classify(const uint64_t *s_rank, int e, int *size, bool *head)
{
    const uint64_t r = s_rank[e + kPcMaxG];
    int left = 0, right = 0;
#pragma unroll
    for (int i = 1; i <= kPcMaxG; ++i)
        if (left == i - 1 && s_rank[e + kPcMaxG - i] == r) left = i;
#pragma unroll
    for (int i = 1; i <= kPcMaxG; ++i)
        if (right == i - 1 && s_rank[e + kPcMaxG + i] == r) right = i;
    const int sz = left + right + 1;
    *size = sz > kPcMaxG ? kPcMaxG + 1 : sz;
    *head = left == 0;
}

// Exclusive prefix sums, in list order (item-major, then wave, then lane), of two per-position counts of a tile.
// Returns through excl_a / excl_b; totals in tot[0..1] (valid after the call for every thread).
struct PcTileScan {
    uint32_t wave_sum[2][kPcItems][kPcThreads / kWave];
    uint32_t tot[2];
};

__device__ __forceinline__ void pc_tile_scan(PcTileScan &sh, const uint32_t (&ca)[kPcItems], const uint32_t (&cb)[kPcItems],
                                             uint32_t (&excl_a)[kPcItems], uint32_t (&excl_b)[kPcItems])
{
    const int lane = lane_id(), wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const uint32_t ia = wave_incl_sum(ca[k]), ib = wave_incl_sum(cb[k]);
        excl_a[k] = ia - ca[k];
        excl_b[k] = ib - cb[k];
        if (lane == kWave - 1) { sh.wave_sum[0][k][wv] = ia; sh.wave_sum[1][k][wv] = ib; }
    }
    __syncthreads();
    if (wv < 2) {
        uint32_t *c = &sh.wave_sum[wv][0][0];
        const uint32_t v = lane < kPcItems * (kPcThreads / kWave) ? c[lane] : 0u;
        const uint32_t incl = wave_incl_sum(v);
        if (lane < kPcItems * (kPcThreads / kWave)) c[lane] = incl - v;
        if (lane == kWave - 1) sh.tot[wv] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        excl_a[k] += sh.wave_sum[0][k][wv];
        excl_b[k] += sh.wave_sum[1][k][wv];
    }
}

__device__ __forceinline__ void pc_load_ranks(uint64_t *s_rank, const uint64_t *__restrict__ rank, int64_t j0, int64_t m)
{
    for (int e = threadIdx.x; e < kPcTile + 2 * kPcMaxG; e += kPcThreads) {
        const int64_t j = j0 - kPcMaxG + e;
        s_rank[e] = (j >= 0 && j < m) ? rank[j] : ~0ull;
    }
}

// One pass over X in tiles of kPcTile entries.  kWrite = false: tile_cnt[2t] = records, [2t+1] = copied entries of
// tile t.  kWrite = true: tile_cnt holds the exclusive prefix sums and the records / copies are written.
template <typename IdxT, bool kWrite>
__global__ __launch_bounds__(kPcThreads) void pair_split_kernel(const uint64_t *__restrict__ rank,
                                                                const IdxT *__restrict__ suf, int64_t m, int ib, int maxg,
                                                                uint32_t *__restrict__ tile_cnt,
                                                                uint64_t *__restrict__ rec_key, IdxT *__restrict__ rec_ord,
                                                                uint64_t *__restrict__ out_rank, IdxT *__restrict__ out_suf)
{
    __shared__ uint64_t s_rank[kPcTile + 2 * kPcMaxG];
    __shared__ PcTileScan sh;
    const int t = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * kPcTile;
    pc_load_ranks(s_rank, rank, j0, m);
    __syncthreads();
    uint32_t npairs[kPcItems], ncopy[kPcItems], op[kPcItems], oc[kPcItems];
    int size[kPcItems];
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        bool head;
        pc_classify(s_rank, e, &size[k], &head);
        const bool valid = j0 + e < m;
        npairs[k] = (valid && head && size[k] >= 2 && size[k] <= maxg) ? (uint32_t)(size[k] * (size[k] - 1) / 2) : 0u;
        ncopy[k] = (valid && size[k] > maxg) ? 1u : 0u;
    }
    pc_tile_scan(sh, npairs, ncopy, op, oc);
    if (!kWrite) {
        if (t < 2) tile_cnt[2 * (int64_t)blockIdx.x + t] = sh.tot[t];
        return;
    }
    const int64_t base_p = tile_cnt[2 * (int64_t)blockIdx.x], base_c = tile_cnt[2 * (int64_t)blockIdx.x + 1];
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        const int64_t j = j0 + e;
        if (npairs[k]) {
            uint64_t s[kPcMaxG];
#pragma unroll
            for (int u = 0; u < kPcMaxG; ++u) s[u] = u < size[k] ? (uint64_t)suf[j + u] : 0ull;
            int64_t o = base_p + op[k];
#pragma unroll
            for (int u = 0; u < kPcMaxG; ++u) {
#pragma unroll
                for (int v = u + 1; v < kPcMaxG; ++v) {
                    if (v < size[k]) {
                        const uint64_t x = s[u] < s[v] ? s[u] : s[v], y = s[u] < s[v] ? s[v] : s[u];
                        rec_key[o] = ((y - x) << ib) | x;
                        rec_ord[o] = maxg == 2 ? (IdxT)s_rank[e + kPcMaxG] : (IdxT)o;     // pairs only: the rank travels along
                        ++o;
                    }
                }
            }
        }
        if (ncopy[k]) {
            const int64_t o = base_c + oc[k];
            out_rank[o] = s_rank[e + kPcMaxG];
            out_suf[o] = suf[j];
        }
    }
}

// tile_cnt[2t], tile_cnt[2t+1] -> exclusive prefix sums over the tiles (two independent sums), totals to ctr.
// One workgroup: every thread sums a contiguous run of tiles, the runs are scanned, the run is rewritten.
static __global__ __launch_bounds__(kPcScanThreads) void pair_scan_kernel(uint32_t *__restrict__ tile_cnt, int64_t ntiles,
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

// status of a chain end: 1 x first, 2 y first, 3 blocked, 4 far (takes the answer of record far[i], further right)
constexpr uint8_t kPcXFirst = 1, kPcYFirst = 2, kPcBlocked = 3, kPcFar = 4;
constexpr uint32_t kPcNone = 0xffffffffu;
constexpr int kPcJumps = 64;              // h-steps a chain end tries through larger tie groups (only chain ends pay for them)
constexpr int kPcResolveRounds = 8;       // pointer-jumping launches: far links nested up to 2^8 deep

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

// Record i = (d, x) is a LINK if record i+1 is (d, x+1).  Otherwise it ends a chain and is evaluated:
//   x' = x+1, y' = y+1 (x and y agree on >= 1 character, so order(x, y) = order(x', y'));
//   y' = n (empty suffix): y first.  ISA[x'] != ISA[y']: decided.  Same tie group (>= h equal characters): if the
//   group is small, (d, x') is a record further right (binary search: the records are sorted) and answers for this
//   chain too -- a FAR link; else step h characters further, a few times; else blocked.
// nt[i] = index of the first chain end at or after i inside the tile (kPcNone: none), tile_head = nt of the
// tile's first record.
// sort_mask: the key bits the records are sorted by -- all of them, or only x when every group is a pair (a text
// position then occurs in one record only, and four digit passes do instead of seven).
template <typename IdxT>
__global__ __launch_bounds__(kPcThreads) void pair_link_kernel(const uint64_t *__restrict__ rec_key, int64_t cnt, int ib,
                                                               uint64_t sort_mask, const IdxT *__restrict__ ISA, int64_t n, int64_t h,
                                                               uint32_t *__restrict__ nt, uint8_t *__restrict__ tstat,
                                                               uint32_t *__restrict__ far, uint32_t *__restrict__ tile_head)
{
    __shared__ uint32_t s_nt[kPcTile];
    __shared__ uint32_t wave_first[kPcThreads / kWave];
    const int t = threadIdx.x, lane = lane_id(), wv = t >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * kPcTile;
    const uint64_t xmask = (1ull << ib) - 1;
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        const int64_t i = i0 + e;
        uint32_t mine = kPcNone;
        if (i < cnt) {
            const uint64_t key = rec_key[i];
            const bool link = i + 1 < cnt && rec_key[i + 1] == key + 1;       // x + 1 < y <= n - 1: no carry into d
            if (!link) {
                mine = (uint32_t)i;
                const uint64_t dpart = key & ~xmask;
                const int64_t d = (int64_t)(key >> ib);
                int64_t x = (int64_t)(key & xmask) + 1;
                uint8_t s = kPcBlocked;
                uint32_t target = 0;
                for (int j = 0; j <= kPcJumps; ++j) {
                    const int64_t y = x + d;
                    if (y >= n) { s = kPcYFirst; break; }             // x < y: y runs out first, the shorter suffix is smaller
                    const int64_t rx = (int64_t)ISA[x], ry = (int64_t)ISA[y];
                    if (rx != ry) { s = rx < ry ? kPcXFirst : kPcYFirst; break; }
                    const uint64_t want = dpart | (uint64_t)x;
                    int64_t lo = i + 1, hi = cnt;                     // lower bound of (d, x) among the records to the right
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if ((rec_key[mid] & sort_mask) < (want & sort_mask)) lo = mid + 1; else hi = mid;
                    }
                    if (lo < cnt && rec_key[lo] == want) { s = kPcFar; target = (uint32_t)lo; break; }
                    x += h;
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
static __global__ __launch_bounds__(kPcScanThreads) void pair_carry_kernel(const uint32_t *__restrict__ tile_head, int64_t ntiles,
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
static __global__ __launch_bounds__(kPcThreads) void pair_resolve_kernel(const uint32_t *__restrict__ nt,
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

// Pairs only (long lists): the record is the whole group and carries its rank, so it is finished right here, in
// RECORD order -- decided: SA[rank], SA[rank+1] and the ISA entry of the second; blocked: both members appended
// behind the copied entries.  (That the blocked pairs come out ordered by text position matters: the doubling rounds
// they return to gather ISA[x+h] and ISA[x+d+h] for consecutive x, i.e. from consecutive addresses -- measured on
// the enwik-like 256 MiB input: 0.2 ms per round for 2e7 entries instead of 0.55 ms in rank order.)
template <typename IdxT>
__global__ __launch_bounds__(kPcThreads) void pair_emit_kernel(const uint64_t *__restrict__ rec_key,
                                                               const IdxT *__restrict__ rec_rank, int64_t cnt, int xbits,
                                                               const uint32_t *__restrict__ nt, const uint32_t *__restrict__ carry,
                                                               const uint8_t *__restrict__ tstat, IdxT *__restrict__ SA,
                                                               IdxT *__restrict__ ISA, uint64_t *__restrict__ out_rank,
                                                               IdxT *__restrict__ out_suf, PairCounters *__restrict__ ctr)
{
    __shared__ uint32_t wave_cnt[kPcThreads / kWave];
    __shared__ unsigned long long s_base;
    const int64_t i = (int64_t)blockIdx.x * kPcThreads + threadIdx.x;
    uint32_t ans = 0;
    int64_t x = 0, y = 0, r = 0;
    if (i < cnt) {
        ans = tstat[pc_chain_end(nt, carry, i)];
        if (ans == kPcFar) ans = kPcBlocked;              // nested deeper than the resolve rounds reach: back to doubling
        const uint64_t key = rec_key[i];
        x = (int64_t)(key & ((1ull << xbits) - 1));
        y = x + (int64_t)(key >> xbits);
        r = (int64_t)rec_rank[i];
    }
    if (ans == kPcXFirst) { SA[r] = (IdxT)x; SA[r + 1] = (IdxT)y; ISA[y] = (IdxT)(r + 1); }
    if (ans == kPcYFirst) { SA[r] = (IdxT)y; SA[r + 1] = (IdxT)x; ISA[x] = (IdxT)(r + 1); }
    const uint64_t blocked = __ballot(ans == kPcBlocked);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 0) wave_cnt[wv] = (uint32_t)__popcll(blocked);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kPcThreads / kWave; ++w) { const uint32_t c = wave_cnt[w]; wave_cnt[w] = tot; tot += c; }
        s_base = tot ? atomicAdd(&ctr->list, 2ull * tot) : 0ull;
    }
    __syncthreads();
    if (ans == kPcBlocked) {
        const int64_t o = (int64_t)s_base + 2 * ((int64_t)wave_cnt[wv] + mask_rank_lt(blocked));
        out_rank[o] = (uint64_t)r; out_suf[o] = (IdxT)x;
        out_rank[o + 1] = (uint64_t)r; out_suf[o + 1] = (IdxT)y;
    }
}

// answer of record i (its chain end's status) -> answer[ordinal]: back in list order for the finish pass
template <typename IdxT>
__global__ __launch_bounds__(kPcThreads) void pair_answer_kernel(const IdxT *__restrict__ rec_ord, int64_t cnt,
                                                                 const uint32_t *__restrict__ nt, const uint32_t *__restrict__ carry,
                                                                 const uint8_t *__restrict__ tstat, uint8_t *__restrict__ answer)
{
    const int64_t i = (int64_t)blockIdx.x * kPcThreads + threadIdx.x;
    if (i >= cnt) return;
    uint8_t a = tstat[pc_chain_end(nt, carry, i)];
    if (a == kPcFar) a = kPcBlocked;                      // nested deeper than the resolve rounds reach: back to doubling
    answer[(int64_t)rec_ord[i]] = a;
}

// Over X again, same tiles and the same record numbering as pair_split_kernel (tile_cnt still holds the prefix
// sums): the head of a small group reads the answers of its pairs.
template <typename IdxT>
__global__ __launch_bounds__(kPcThreads) void pair_finish_kernel(const uint64_t *__restrict__ rank, const IdxT *__restrict__ suf,
                                                                 int64_t m, int maxg, const uint32_t *__restrict__ tile_cnt,
                                                                 const uint8_t *__restrict__ answer, IdxT *__restrict__ SA,
                                                                 IdxT *__restrict__ ISA, uint64_t *__restrict__ out_rank,
                                                                 IdxT *__restrict__ out_suf, PairCounters *__restrict__ ctr)
{
    __shared__ uint64_t s_rank[kPcTile + 2 * kPcMaxG];
    __shared__ PcTileScan sh;
    __shared__ unsigned long long s_base;
    const int t = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * kPcTile;
    pc_load_ranks(s_rank, rank, j0, m);
    __syncthreads();
    uint32_t npairs[kPcItems], nkeep[kPcItems], op[kPcItems], ok[kPcItems], zero[kPcItems];
    int size[kPcItems];
    const int64_t base_p = tile_cnt[2 * (int64_t)blockIdx.x];
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        const int e = k * kPcThreads + t;
        bool head;
        pc_classify(s_rank, e, &size[k], &head);
        const bool valid = j0 + e < m;
        npairs[k] = (valid && head && size[k] >= 2 && size[k] <= maxg) ? (uint32_t)(size[k] * (size[k] - 1) / 2) : 0u;
        nkeep[k] = 0;
        zero[k] = 0;
    }
    pc_tile_scan(sh, npairs, zero, op, ok);               // the record numbering of the split pass
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        if (npairs[k]) {
            const int64_t j = j0 + k * kPcThreads + t;
            uint64_t s[kPcMaxG];
            int place[kPcMaxG];
#pragma unroll
            for (int u = 0; u < kPcMaxG; ++u) { s[u] = u < size[k] ? (uint64_t)suf[j + u] : 0ull; place[u] = 0; }
            int64_t o = base_p + op[k];
            bool all = true;
#pragma unroll
            for (int u = 0; u < kPcMaxG; ++u) {
#pragma unroll
                for (int v = u + 1; v < kPcMaxG; ++v) {
                    if (v < size[k]) {
                        const uint8_t a = answer[o++];
                        // the record was (min, max) of the two text positions: "x first" = the smaller position first
                        const bool u_first = (a == kPcXFirst) == (s[u] < s[v]);
                        if (a != kPcXFirst && a != kPcYFirst) all = false;
                        place[u] += u_first ? 0 : 1;
                        place[v] += u_first ? 1 : 0;
                    }
                }
            }
            if (all) {
                const int64_t r = (int64_t)s_rank[k * kPcThreads + t + kPcMaxG];
#pragma unroll
                for (int u = 0; u < kPcMaxG; ++u) {
                    if (u < size[k]) {
                        SA[r + place[u]] = (IdxT)s[u];
                        if (place[u]) ISA[(int64_t)s[u]] = (IdxT)(r + place[u]);
                    }
                }
            } else {
                nkeep[k] = (uint32_t)size[k];
            }
        }
    }
    __syncthreads();                                      // (the scan's shared state is reused)
    pc_tile_scan(sh, nkeep, zero, op, ok);
    if (t == 0) s_base = sh.tot[0] ? atomicAdd(&ctr->list, (unsigned long long)sh.tot[0]) : 0ull;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPcItems; ++k) {
        if (nkeep[k]) {
            const int64_t j = j0 + k * kPcThreads + t;
            const int64_t o = (int64_t)s_base + op[k];
            const uint64_t r = s_rank[k * kPcThreads + t + kPcMaxG];
#pragma unroll
            for (int u = 0; u < kPcMaxG; ++u) {
                if (u < size[k]) { out_rank[o + u] = r; out_suf[o + u] = suf[j + u]; }
            }
        }
    }
}

}  // namespace dq
