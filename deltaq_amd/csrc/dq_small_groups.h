// dq_small_groups.h -- one doubling round for the tie groups of <= 8 suffixes, in a single pass.
//
// After the first rounds most tie groups are tiny (pairs inside long repeats), and the round
// is nothing but random rank gathers and scatters; pushing those elements through 6-8 radix
// passes wastes ~200 B of traffic per element.  small_group_round_kernel instead finishes a
// round for every group of <= kSgMaxG members in one pass over the tied list X:
//
//   load      one lane per element: (rank, suffix) coalesced, key2 = ISA[s+h]+h | n-1-s gathered
//             by the lane itself (one outstanding random read per lane), parked in LDS
//   classify  each lane scans <= 8 LDS neighbours on both sides for its group's extent
//   sort      a lane's place inside its group = #(members with a smaller key2) + #(equal ones
//             before it): <= 8 LDS reads, no divergent sorting network
//   exchange  the record moves to the LDS slot of its sorted position, so the workgroup now
//             holds its part of X in final order: SA writes are contiguous runs and the
//             survivors of one subgroup are adjacent
//   emit      resolved    -> SA[rank + place] = s (final)
//             still tied  -> appended to the next list T as (new rank, s)
//             rank moved  -> appended to the update list U as (s, new rank)
//             group > 8   -> appended to L as (rank << kbits | key2, s) for the radix path
//
// The inverse suffix array must not change while other workgroups still gather from it, so
// rank updates are deferred to isa_update_kernel (the next launch).  Lists are appended with
// one atomic per workgroup and list: their order across workgroups is arbitrary, which is
// fine -- the only invariant the next round needs is "members of a group are adjacent", and
// a group is always emitted by the one workgroup that owns its first member.
#pragma once
#include <type_traits>
#include "dq_device_utils.h"

namespace dq {

constexpr int kSgMaxG = 8;                          // long lists: the LDS work per entry grows with the group cap
constexpr int kSgMaxGShort = 32;                    // short lists are launch-bound: take larger groups too, so that
                                                    // the radix side path (7 launches + a host round trip) dies out sooner
constexpr int kSgThreads = 512;
constexpr int kSgItems = 8;
constexpr int kSgWaves = kSgThreads / kWave;
constexpr int kSgSpan = kSgThreads * kSgItems;      // list positions a workgroup looks at
template <int kMaxG> constexpr int sg_tile() { return kSgSpan - kMaxG; }   // ... of which it owns the groups starting there

// Appended-entry counters.  One same-address atomic costs ~11 ns on MI355X (they serialise in
// one L2 channel), so a workgroup reserves its T and U space with ONE packed atomic
// (T count in the low half, U count in the high half); L is touched only where large groups exist.
struct SmallGroupCounters {
    unsigned long long tied_moved;  // low 32 bits: entries appended to T, high 32 bits: to U
    unsigned long long large;       // entries appended to L
};

template <typename IdxT, int kMaxG = kSgMaxG>
__global__ __launch_bounds__(kSgThreads) void small_group_round_kernel(
    const uint64_t *__restrict__ rank, const IdxT *__restrict__ suf, const IdxT *__restrict__ ISA,
    int64_t m, int64_t n, int64_t h, int kbits, IdxT *__restrict__ SA,
    uint64_t *__restrict__ t_rank, IdxT *__restrict__ t_suf,
    uint64_t *__restrict__ l_key, IdxT *__restrict__ l_suf,
    uint64_t *__restrict__ u_rank_end, IdxT *__restrict__ u_suf_end,      // U grows DOWNWARD from these
    SmallGroupCounters *__restrict__ ctr, const SmallGroupCounters *__restrict__ prev = nullptr)
{
    // chained rounds (no host round trip in between): the list length is what the previous round appended
    // to T; m is then only the bound the grid was sized for
    if (prev) {
        const int64_t real = (int64_t)(prev->tied_moved & 0xffffffffull);
        m = real < m ? real : m;
    }
    constexpr int kTile = sg_tile<kMaxG>();
    if ((int64_t)blockIdx.x * kTile >= m) return;
    // ranks and key2 values are < n + h <= 2n: unsigned 32 bits are enough for the int32 index type
    using ElemT = typename std::make_unsigned<IdxT>::type;
    constexpr int kHalo = kMaxG;
    constexpr ElemT kNone = ~(ElemT)0;
    __shared__ ElemT s_rank[kSgSpan + 2 * kHalo];      // phase 2 on: slot_dest
    __shared__ ElemT s_key2[kSgSpan + 2 * kHalo];      // phase 2 on: slot_rank
    __shared__ IdxT slot_suf[kSgSpan];
    __shared__ uint8_t slot_flag[kSgSpan];             // 0 empty, bit0 present, bit1 tied, bit2 moved
    __shared__ uint32_t wave_cnt[3][kSgItems][kSgWaves];
    __shared__ unsigned long long base[3];
    ElemT *slot_dest = s_rank;
    ElemT *slot_rank = s_key2;

    const int t = threadIdx.x;
    const int lane = lane_id();
    const int wv = t >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * kTile;

    ElemT r[kSgItems], k2[kSgItems];
    IdxT s[kSgItems];
#pragma unroll
    for (int k = 0; k < kSgItems; ++k) {
        const int e = k * kSgThreads + t;
        const int64_t j = j0 + e;
        r[k] = kNone; s[k] = 0;
        if (j < m) { r[k] = (ElemT)rank[j]; s[k] = suf[j]; }
    }
#pragma unroll
    for (int k = 0; k < kSgItems; ++k) {
        const int64_t q = (int64_t)s[k] + h;
        k2[k] = 0;
        if (r[k] != kNone)
            k2[k] = q < n ? (ElemT)((int64_t)ISA[q] + h) : (ElemT)(n - 1 - (int64_t)s[k]);   // as gather_key2_kernel
    }
#pragma unroll
    for (int k = 0; k < kSgItems; ++k) {
        const int e = k * kSgThreads + t;
        s_rank[kHalo + e] = r[k];
        s_key2[kHalo + e] = k2[k];
        slot_flag[e] = 0;
    }
    if (t < kHalo) {
        const int64_t jl = j0 - kHalo + t;
        s_rank[t] = jl >= 0 ? (ElemT)rank[jl] : kNone;
        const int64_t jr = j0 + kSgSpan + t;
        s_rank[kHalo + kSgSpan + t] = jr < m ? (ElemT)rank[jr] : kNone;
    }
    __syncthreads();

    int slot[kSgItems];                                 // -1: not a small-group element this workgroup owns
    uint8_t flag[kSgItems];
    ElemT nrank[kSgItems], dest[kSgItems];
    bool own_large[kSgItems];
#pragma unroll
    for (int k = 0; k < kSgItems; ++k) {
        const int e = k * kSgThreads + t;
        const bool valid = r[k] != kNone;
        // extent of the group: equal ranks to the left / right, each capped at kMaxG
        int left = 0, right = 0;
#pragma unroll
        for (int i = 1; i <= kMaxG; ++i)
            if (left == i - 1 && s_rank[kHalo + e - i] == r[k]) left = i;
#pragma unroll
        for (int i = 1; i <= kMaxG; ++i)
            if (right == i - 1 && s_rank[kHalo + e + i] == r[k]) right = i;
        const bool large = valid && (left + right + 1 > kMaxG);
        const int head = e - left;                      // span index of the group's first member
        const bool own_small = valid && !large && head >= 0 && head < kTile;
        own_large[k] = large && e < kTile;
        slot[k] = -1; flag[k] = 0; nrank[k] = 0; dest[k] = 0;
        if (own_small) {
            int less = 0, eq = 0, eq_before = 0;
#pragma unroll
            for (int i = -kMaxG + 1; i < kMaxG; ++i) {
                if (i >= -left && i <= right) {
                    const ElemT o = s_key2[kHalo + e + i];
                    less += o < k2[k];
                    eq += o == k2[k];
                    eq_before += (o == k2[k]) && i < 0;
                }
            }
            slot[k] = head + less + eq_before;          // <= kTile - 1 + kMaxG - 1 < kSgSpan
            nrank[k] = r[k] + (ElemT)less;
            dest[k] = r[k] + (ElemT)(less + eq_before);
            flag[k] = (uint8_t)(1 | (eq > 1 ? 2 : 0) | (less != 0 ? 4 : 0));
        }
    }
    __syncthreads();                                    // every reader of s_rank / s_key2 is done
#pragma unroll
    for (int k = 0; k < kSgItems; ++k) {
        if (slot[k] >= 0) {
            slot_dest[slot[k]] = dest[k];
            slot_rank[slot[k]] = nrank[k];
            slot_suf[slot[k]] = s[k];
            slot_flag[slot[k]] = flag[k];
        }
    }
    __syncthreads();

    // from here on item k of lane t speaks for the record in sorted slot k*kSgThreads + t
    uint32_t f[kSgItems];
    uint64_t bt[kSgItems], bl[kSgItems], bu[kSgItems];
#pragma unroll
    for (int k = 0; k < kSgItems; ++k) {
        const int e = k * kSgThreads + t;
        f[k] = slot_flag[e];
        if ((f[k] & 3) == 1) SA[slot_dest[e]] = slot_suf[e];      // resolved now; tied ones are written when they resolve
        bt[k] = __ballot((f[k] & 2) != 0);
        bl[k] = __ballot(own_large[k]);
        bu[k] = __ballot((f[k] & 4) != 0);
        if (lane == 0) {
            wave_cnt[0][k][wv] = (uint32_t)__popcll(bt[k]);
            wave_cnt[1][k][wv] = (uint32_t)__popcll(bl[k]);
            wave_cnt[2][k][wv] = (uint32_t)__popcll(bu[k]);
        }
    }
    __syncthreads();
    // exclusive scan of the 3 x 64 (item, wave) counts in emission order: one wave per list
    if (wv < 3) {
        uint32_t *cnts = &wave_cnt[wv][0][0];
        const uint32_t c = cnts[lane];
        const uint32_t incl = wave_incl_sum(c);
        cnts[lane] = incl - c;
        const uint32_t tot = __shfl(incl, kWave - 1, kWave);
        if (lane == 0) base[wv] = tot;                 // totals first; bases below
    }
    __syncthreads();
    if (t == 0) {
        const unsigned long long nt = base[0], nl = base[1], nu = base[2];
        unsigned long long tu = 0;
        if (nt | nu) tu = atomicAdd(&ctr->tied_moved, nt | (nu << 32));
        base[0] = tu & 0xffffffffull;
        base[2] = tu >> 32;
        base[1] = nl ? atomicAdd(&ctr->large, nl) : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSgItems; ++k) {
        const int e = k * kSgThreads + t;
        if (f[k] & 2) {
            const int64_t p = (int64_t)base[0] + wave_cnt[0][k][wv] + mask_rank_lt(bt[k]);
            t_rank[p] = (uint64_t)slot_rank[e];
            t_suf[p] = slot_suf[e];
        }
        if (own_large[k]) {
            const int64_t p = (int64_t)base[1] + wave_cnt[1][k][wv] + mask_rank_lt(bl[k]);
            l_key[p] = ((uint64_t)r[k] << kbits) | (uint64_t)k2[k];
            l_suf[p] = s[k];
        }
        if (f[k] & 4) {
            const int64_t p = (int64_t)base[2] + wave_cnt[2][k][wv] + mask_rank_lt(bu[k]);
            u_rank_end[-1 - p] = (uint64_t)slot_rank[e];
            u_suf_end[-1 - p] = slot_suf[e];
        }
    }
}

// ISA[s] = new rank for the entries of the update list (stored downward from the *_end pointers).
// u_ib > 0: the entries are single words (rank << u_ib | suffix) in the rank array (mid_group_round_kernel writes them
// that way when two indices fit one word), the suffix array is not used.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void isa_update_kernel(const uint64_t *__restrict__ u_rank_end,
                                                            const IdxT *__restrict__ u_suf_end, int64_t count,
                                                            IdxT *__restrict__ ISA,
                                                            const SmallGroupCounters *__restrict__ cnt_dev = nullptr,
                                                            int u_ib = 0)
{
    if (cnt_dev) count = (int64_t)(cnt_dev->tied_moved >> 32);            // chained rounds: length from the device
    if (u_ib > 0) {
        const uint64_t mask = (1ull << u_ib) - 1;
        for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < count; p += (int64_t)gridDim.x * kBlock) {
            const uint64_t wd = u_rank_end[-1 - p];
            ISA[wd & mask] = (IdxT)(wd >> u_ib);
        }
        return;
    }
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < count; p += (int64_t)gridDim.x * kBlock)
        ISA[u_suf_end[-1 - p]] = (IdxT)u_rank_end[-1 - p];
}

// The same for update words that have been BINNED by the top bits of the suffix first (two word passes of the radix
// sorter): consecutive words then fall into one window of 2^(ib-16) suffixes -- the 4-byte writes of a workgroup
// land in a few KB of the array and leave its L2 as whole lines instead of one read-modify-write per entry.
// The all-ones word is padding (an alignment filler in front of the list) and is skipped: a real word has bit 63 clear
// (2 * ib <= 63), while the filler's suffix field reads n - 1 -- a real suffix -- when n is a power of two.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void isa_update_words_kernel(const uint64_t *__restrict__ words, int64_t count,
                                                                  int ib, int64_t n, IdxT *__restrict__ ISA)
{
    const uint64_t mask = (1ull << ib) - 1;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < count; p += (int64_t)gridDim.x * kBlock) {
        const uint64_t wd = words[p];
        const uint64_t sfx = wd & mask;
        if (wd != ~0ull && (int64_t)sfx < n) ISA[sfx] = (IdxT)(wd >> ib);
    }
}

// A DOUBLED text (block + block, as bzip2's rotation sort asks for: dq_bz2.h) ties suffix i with suffix i + half for
// half - i characters -- the shorter one is a prefix of the longer -- although their order is known from the start: the
// shorter one first.  Prefix doubling would carry all these pairs to h > half (17 rounds over the whole list for a
// 900 kB block of random bytes whose real ties end after the first).  So between the rounds every tie group that is
// exactly such a pair is written down -- its two slots of the suffix array and its two final ranks (ranks finer than
// the current depth asks for are always welcome) -- and leaves the list; the rounds go on with the real ties only.
constexpr int kTwThreads = 256, kTwItems = 8;
constexpr int kTwTile = kTwThreads * kTwItems;          // (= the tile the pair-chain counters are sized for)

// list position j: 0 = stays (its group is not such a pair), 1 / 2 = first / second entry of a pair.  *head: first of its group.
template <typename IdxT>
__device__ __forceinline__ int twin_class(const uint64_t *__restrict__ rank, const uint32_t *__restrict__ rank32,
                                          const IdxT *__restrict__ suf, int64_t m, int64_t half, int64_t j, bool *head)
{
    auto rk = [&](int64_t q) -> uint64_t { return rank32 ? (uint64_t)rank32[q] : rank[q]; };
    const uint64_t r = rk(j);
    *head = j == 0 || rk(j - 1) != r;
    const int64_t g0 = *head ? j : j - 1;                // where the pair would begin
    if (!*head && g0 > 0 && rk(g0 - 1) == r) return 0;   // third or later member
    if (g0 + 1 >= m || rk(g0 + 1) != r) return 0;
    if (g0 + 2 < m && rk(g0 + 2) == r) return 0;
    const int64_t a = (int64_t)suf[g0], b = (int64_t)suf[g0 + 1];
    if (a - b != half && b - a != half) return 0;
    return *head ? 1 : 2;
}

// pass 1: write the pairs down; per tile, the groups and the entries that stay (tile_cnt[2t], [2t + 1])
template <typename IdxT>
__global__ __launch_bounds__(kTwThreads) void twin_mark_kernel(const uint64_t *__restrict__ rank, const uint32_t *__restrict__ rank32,
                                                               const IdxT *__restrict__ suf, int64_t m, int64_t half,
                                                               IdxT *__restrict__ SA, IdxT *__restrict__ ISA, uint32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t part[2][kTwThreads / kWave];
    const int64_t j0 = (int64_t)blockIdx.x * kTwTile + (int64_t)threadIdx.x * kTwItems;
    uint32_t kept = 0, groups = 0;
    for (int k = 0; k < kTwItems; ++k) {
        const int64_t j = j0 + k;
        if (j >= m) break;
        bool head;
        const int cls = twin_class<IdxT>(rank, rank32, suf, m, half, j, &head);
        if (cls == 0) { ++kept; groups += head ? 1u : 0u; }
        else if (cls == 1) {
            const uint64_t r = rank32 ? (uint64_t)rank32[j] : rank[j];
            const IdxT a = suf[j], b = suf[j + 1];
            const IdxT hi = a > b ? a : b, lo = a > b ? b : a;
            SA[r] = hi; SA[r + 1] = lo;
            ISA[hi] = (IdxT)r; ISA[lo] = (IdxT)(r + 1);
        }
    }
    const uint32_t ik = wave_incl_sum(kept), ig = wave_incl_sum(groups);
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (lane == kWave - 1) { part[0][wv] = ig; part[1][wv] = ik; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t g = 0, kp = 0;
        for (int i = 0; i < kTwThreads / kWave; ++i) { g += part[0][i]; kp += part[1][i]; }
        tile_cnt[2 * blockIdx.x] = g;
        tile_cnt[2 * blockIdx.x + 1] = kp;
    }
}

// pass 2 (tile_cnt scanned by pair_scan_kernel): the entries that stay, in order
template <typename IdxT>
__global__ __launch_bounds__(kTwThreads) void twin_compact_kernel(const uint64_t *__restrict__ rank, const uint32_t *__restrict__ rank32,
                                                                  const IdxT *__restrict__ suf, int64_t m, int64_t half,
                                                                  const uint32_t *__restrict__ tile_cnt,
                                                                  uint64_t *__restrict__ out_rank, IdxT *__restrict__ out_suf)
{
    __shared__ uint32_t part[kTwThreads / kWave];
    const int64_t j0 = (int64_t)blockIdx.x * kTwTile + (int64_t)threadIdx.x * kTwItems;
    uint32_t stays = 0, kept = 0;
    for (int k = 0; k < kTwItems; ++k) {
        const int64_t j = j0 + k;
        if (j >= m) break;
        bool head;
        if (twin_class<IdxT>(rank, rank32, suf, m, half, j, &head) == 0) { stays |= 1u << k; ++kept; }
    }
    const uint32_t incl = wave_incl_sum(kept);
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    if (lane == kWave - 1) part[wv] = incl;
    __syncthreads();
    int64_t p = (int64_t)tile_cnt[2 * blockIdx.x + 1] + (incl - kept);
    for (int i = 0; i < wv; ++i) p += part[i];
    for (int k = 0; k < kTwItems; ++k) {
        if (!((stays >> k) & 1u)) continue;
        const int64_t j = j0 + k;
        out_rank[p] = rank32 ? (uint64_t)rank32[j] : rank[j];
        out_suf[p] = suf[j];
        ++p;
    }
}

}  // namespace dq
