// dq_mid_groups.h -- one doubling round for the tie groups of up to kG suffixes (kG = 256 ... 1024), in a single
// pass, inside LDS: the middle size class between small_group_round_kernel (groups <= 8, dq_small_groups.h) and the
// global radix passes (8 passes of 24 B per entry for a 57-bit composite key).
//
// Text-like inputs leave the first doubling rounds with a heavy-tailed group size distribution (256 MiB of
// enwik-style text after the coded round 0: half of the tied suffixes sit in groups of more than 8, nine tenths of
// those in groups of at most a few thousand).  The members of a group are adjacent in the rank-ordered list X, so a
// workgroup that holds a stretch of X in LDS can finish every group that lies inside it:
//
//   load      ranks of kG + kSpan list positions (left halo: ranks only), suffixes of the kSpan span
//   heads     head[c] = rank[c] != rank[c-1]; block-wide max-scan -> every position knows its group's first
//             position; every head closes the group before it -> group sizes.  A group belongs to the workgroup
//             whose tile [0, kTile) holds its head; with size <= kG it lies inside the span [0, kTile + kG).
//             A group that is longer, or whose head lies more than kG to the left, is "large": each workgroup
//             appends ITS OWN tile's members of large groups to the radix list L.  Both sides of a tile boundary
//             see the true size of every group of <= kG members (left halo / right overhang), so they agree.
//   gather    key2 = ISA[s + h] + h | n - 1 - s only for the members that need it (owned groups, own large members)
//   place     member's place inside its group = #(smaller key2) + #(equal key2 before it): one walk over the group's
//             key2 values in LDS (lanes of a wave mostly share a group: broadcast reads)
//   exchange / emit   exactly as small_group_round_kernel: resolved -> SA, still tied -> T, rank moved -> U,
//             member of a large group -> L as (rank << kbits | key2, s)
//
// Cost per member ~ its group size (the walk), so kG trades LDS work against radix passes; the host picks this
// kernel only while a round still has many groups beyond 8 (the first rounds of a text-like input).
#pragma once
#include <type_traits>
#include "dq_device_utils.h"
#include "dq_small_groups.h"
#include "dq_runs.h"

namespace dq {

constexpr int kMgThreads = 1024;
constexpr int kMgItems = 4;
constexpr int kMgSpan = kMgThreads * kMgItems;          // list positions whose members a workgroup can place
constexpr int kMgWaves = kMgThreads / kWave;
template <int kG> constexpr int mg_tile() { return kMgSpan - kG; }      // heads it owns

// kSteps > 1 (round 5; short lists whose groups all fit the class -- the chained rounds): a member's key is the TUPLE
// (ISA[s + h], ISA[s + 2h], ... ISA[s + kSteps h]) -- ranks at depth >= h of the next kSteps stretches of h bytes -- so a round
// compares (kSteps + 1) h bytes and the depth grows (kSteps + 1)-fold instead of doubling: a repeat of L bytes keeps its
// suffixes tied for log_{kSteps+1}(L / h) rounds instead of log2.  The rounds of a short list are latency, not traffic
// (20 us + 7 us of launches each whatever the list's length), so three gathers per member instead of one are free there;
// long lists are bound by the sectors their gathers move and keep kSteps = 1.  A suffix that ends inside
// stretch j has (n - 1 - s) - j h there -- < h, below every in-range value, the shorter suffix first -- and 0 behind it.
// (int32, kSteps = 1: 57-62 KB of LDS and <= 64 VGPRs, so that two workgroups share a CU -- twice the random gathers in flight)
template <typename IdxT, int kG, int kSteps = 1>
__global__ __launch_bounds__(kMgThreads, (sizeof(IdxT) == 4 && kSteps == 1) ? 8 : 4) void mid_group_round_kernel(
    const uint64_t *__restrict__ rank, const IdxT *__restrict__ suf, const IdxT *__restrict__ ISA,
    int64_t m, int64_t n, int64_t h, int kbits, IdxT *__restrict__ SA,
    uint64_t *__restrict__ t_rank, IdxT *__restrict__ t_suf,
    uint64_t *__restrict__ l_key, IdxT *__restrict__ l_suf,
    uint64_t *__restrict__ u_rank_end, IdxT *__restrict__ u_suf_end,      // U grows DOWNWARD from these
    SmallGroupCounters *__restrict__ ctr, const SmallGroupCounters *__restrict__ prev = nullptr,
    const uint32_t *__restrict__ RL = nullptr /* run lengths of the text (dq_runs.h), or none */,
    const uint8_t *__restrict__ text = nullptr, int run_order = 0 /* 1: this is the run-order round */,
    const uint32_t *__restrict__ rank32 = nullptr /* the ranks as 32-bit values instead of `rank` (first round) */,
    int u_ib = 0 /* > 0: update entries as single words (rank << u_ib | suffix) in u_rank_end, u_suf_end unused */,
    int l_shift = 0 /* > 0: the radix list's keys carry rank >> l_shift as their sort field, the low rank bits as payload */)
{
    // chained rounds (no host round trip in between): the list length is what the previous round appended to T
    if (prev) {
        const int64_t real = (int64_t)(prev->tied_moved & 0xffffffffull);
        m = real < m ? real : m;
    }
    constexpr int kTile = mg_tile<kG>();
    constexpr int kScan = kG + kMgSpan;                 // scan coordinates c = e + kG, e = span position
    constexpr int kPer = (kScan + kMgThreads - 1) / kMgThreads;       // consecutive scan positions per thread
    using ElemT = typename std::make_unsigned<IdxT>::type;
    constexpr ElemT kNone = ~(ElemT)0;
    constexpr uint16_t kOpen = 0xffff;                  // group not closed inside the scan region
    static_assert(kScan < 0x7fff, "head positions are 16-bit");

    __shared__ ElemT s_rank[kScan + 1];                 // [kScan] = the closer behind the span; later: slot_dest
    __shared__ ElemT s_key2[kSteps][kMgSpan];           // [0] later: slot_rank
    constexpr int kMixBytes = 4 * kScan > kMgSpan * (int)sizeof(IdxT) ? 4 * kScan : kMgSpan * (int)sizeof(IdxT);
    __shared__ __attribute__((aligned(16))) char s_mix[kMixBytes];                  // s_head + s_gsize, later slot_suf
    __shared__ uint8_t slot_flag[kMgSpan];
    __shared__ int s_wmax[kMgWaves];
    __shared__ uint32_t wave_cnt[3][kMgItems][kMgWaves];
    __shared__ unsigned long long base[3];
    int16_t *s_head = reinterpret_cast<int16_t *>(s_mix);              // [kScan] first position of the group, -1 unknown
    uint16_t *s_gsize = reinterpret_cast<uint16_t *>(s_mix) + kScan;   // [kScan] by head position
    IdxT *slot_suf = reinterpret_cast<IdxT *>(s_mix);
    ElemT *slot_dest = s_rank;
    ElemT *slot_rank = s_key2[0];

    const int t = threadIdx.x;
    const int lane = lane_id();
    const int wv = t >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * kTile;     // list position of span position 0
    if (j0 >= m) return;

    // ---- load: ranks of the scan region (striped, coalesced), suffixes of the span ----
    for (int c = t; c < kScan + 1; c += kMgThreads) {
        const int64_t j = j0 - kG + c;
        s_rank[c] = (j >= 0 && j < m) ? (rank32 ? (ElemT)rank32[j] : (ElemT)rank[j]) : kNone;
    }
    IdxT s[kMgItems];
#pragma unroll
    for (int k = 0; k < kMgItems; ++k) {
        const int64_t j = j0 + k * kMgThreads + t;
        s[k] = j < m ? suf[j] : (IdxT)0;
        slot_flag[k * kMgThreads + t] = 0;
    }
    __syncthreads();

    // ---- heads: blocked, kPer consecutive scan positions per thread ----
    {
        const int c0 = t * kPer;
        int hp[kPer];
        int run = -1;
        ElemT prev = c0 > 0 && c0 <= kScan ? s_rank[c0 - 1] : kNone;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int c = c0 + i;
            const ElemT r = c < kScan ? s_rank[c] : kNone;
            // (position 0 counts as a head only at the start of the list: a group cut by the left edge of the scan
            // region has its head further left -- unknown, which makes it large if it reaches this tile)
            const bool head = c < kScan && (c == 0 ? (j0 - kG <= 0) : r != prev);
            run = head ? c : run;
            hp[i] = run;
            prev = r;
        }
        // carry in: the last head of the threads before this one
        const int wincl = wave_incl_max(run);
        if (lane == kWave - 1) s_wmax[wv] = wincl;
        int carry = __shfl_up(wincl, 1, kWave);
        if (lane == 0) carry = -1;
        __syncthreads();
        for (int i = 0; i < wv; ++i) carry = s_wmax[i] > carry ? s_wmax[i] : carry;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int c = c0 + i;
            if (c < kScan) {
                const int head_of_prev = i == 0 ? carry : (hp[i - 1] > carry ? hp[i - 1] : carry);
                const int mine = hp[i] > carry ? hp[i] : carry;
                s_head[c] = (int16_t)mine;
                if (mine == c && c > 0 && head_of_prev >= 0) s_gsize[head_of_prev] = (uint16_t)(c - head_of_prev);   // c closes it
                if (c == kScan - 1 && mine >= 0)                     // the last group: closed iff the closer differs
                    s_gsize[mine] = s_rank[kScan] != s_rank[c] ? (uint16_t)(kScan - mine) : kOpen;
            }
        }
    }
    __syncthreads();

    // ---- classify the span positions (striped from here on: e = k * kMgThreads + t) ----
    ElemT r[kMgItems], k2[kMgItems];
    ElemT kx[kMgItems][kSteps > 1 ? kSteps - 1 : 1];  // the further elements of the key tuple (kSteps > 1)
    int ghead[kMgItems], gsize[kMgItems];               // span position of the group's head, its size (owned groups)
    bool own[kMgItems], own_large[kMgItems];
#pragma unroll
    for (int k = 0; k < kMgItems; ++k) {
        const int e = k * kMgThreads + t;
        r[k] = s_rank[kG + e];
        const bool valid = r[k] != kNone;
        const int hc = s_head[kG + e];                  // scan coordinates
        const uint16_t gs = hc >= 0 ? s_gsize[hc] : kOpen;
        const bool large = hc < 0 || gs == kOpen || gs > kG;
        ghead[k] = hc - kG;
        gsize[k] = gs;
        own[k] = valid && !large && ghead[k] >= 0 && ghead[k] < kTile;
        own_large[k] = valid && large && e < kTile;
    }
    // ---- gather key2 where it is needed ----
#pragma unroll
    for (int k = 0; k < kMgItems; ++k) {
        k2[k] = 0;
#pragma unroll
        for (int j = 1; j < kSteps; ++j) kx[k][j - 1] = 0;
        if (own[k] || own_large[k]) {
            // (dq_runs.h: a member that starts with a run of >= h equal bytes takes the rank behind its run; in the
            // run-order round its key is the run's own order, and everybody else's 0 -- no split)
            int64_t off = h;
            bool keyed = false;
            if (RL) {
                const uint32_t r = RL[s[k]];
                if (run_order) { k2[k] = (int64_t)r >= h ? (ElemT)run_order_key(text, n, (int64_t)s[k], r, run_order) : (ElemT)0; keyed = true; }
                else if ((int64_t)r > h) off = (int64_t)r;
            }
            if (!keyed) {
                const int64_t q = (int64_t)s[k] + off;
                k2[k] = q < n ? (ElemT)((int64_t)ISA[q] + h) : (off > h ? (ElemT)0 : (ElemT)(n - 1 - (int64_t)s[k]));   // as gather_key2_kernel
                if (kSteps > 1) {
                    // (no run lengths on this path: the host takes kSteps > 1 only without them)
#pragma unroll
                    for (int j = 1; j < kSteps; ++j) {
                        const int64_t qj = (int64_t)s[k] + (int64_t)(j + 1) * h, qp = (int64_t)s[k] + (int64_t)j * h;
                        kx[k][j - 1] = qj < n ? (ElemT)((int64_t)ISA[qj] + h) : (qp < n ? (ElemT)(n - 1 - (int64_t)s[k] - (int64_t)j * h) : (ElemT)0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kMgItems; ++k) {
        s_key2[0][k * kMgThreads + t] = k2[k];
        if (kSteps > 1) {
#pragma unroll
            for (int j = 1; j < kSteps; ++j) s_key2[j][k * kMgThreads + t] = kx[k][j - 1];
        }
    }
    __syncthreads();

    // ---- place: walk the group ----
    int slot[kMgItems];
    uint8_t flag[kMgItems];
    ElemT nrank[kMgItems], dest[kMgItems];
#pragma unroll
    for (int k = 0; k < kMgItems; ++k) {
        const int e = k * kMgThreads + t;
        slot[k] = -1; flag[k] = 0; nrank[k] = 0; dest[k] = 0;
        // (the loop bound is made wave-uniform per item by the compiler's divergence handling; lanes of a wave hold
        // consecutive positions, so they mostly walk the same group)
        if (own[k]) {
            int less = 0, eq = 0, eq_before = 0;
            const int g0 = ghead[k], me = e - g0;
            for (int i = 0; i < gsize[k]; ++i) {
                const ElemT o = s_key2[0][g0 + i];
                bool lt = o < k2[k], same = o == k2[k];
                if (kSteps > 1) {
#pragma unroll
                    for (int j = 1; j < kSteps; ++j) {
                        const ElemT oj = s_key2[j][g0 + i];
                        lt = lt || (same && oj < kx[k][j - 1]);
                        same = same && oj == kx[k][j - 1];
                    }
                }
                less += lt;
                eq += same;
                eq_before += same && i < me;
            }
            slot[k] = g0 + less + eq_before;
            nrank[k] = r[k] + (ElemT)less;
            dest[k] = r[k] + (ElemT)(less + eq_before);
            flag[k] = (uint8_t)(1 | (eq > 1 ? 2 : 0) | (less != 0 ? 4 : 0));
        }
    }
    __syncthreads();                                    // every reader of s_rank / s_key2 / s_head / s_gsize is done
#pragma unroll
    for (int k = 0; k < kMgItems; ++k) {
        if (slot[k] >= 0) {
            slot_dest[slot[k]] = dest[k];
            slot_rank[slot[k]] = nrank[k];
            slot_suf[slot[k]] = s[k];
            slot_flag[slot[k]] = flag[k];
        }
    }
    __syncthreads();

    // ---- emit: item k of lane t speaks for the record in sorted slot k * kMgThreads + t ----
    uint32_t f[kMgItems];
    uint64_t bt[kMgItems], bl[kMgItems], bu[kMgItems];
#pragma unroll
    for (int k = 0; k < kMgItems; ++k) {
        const int e = k * kMgThreads + t;
        f[k] = slot_flag[e];
        if ((f[k] & 3) == 1) SA[slot_dest[e]] = slot_suf[e];      // resolved now
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
    // exclusive scan of the 3 x (item, wave) counts in emission order: one wave per list (kMgItems * kMgWaves = 64)
    static_assert(kMgItems * kMgWaves == kWave, "one lane per (item, wave) count");
    if (wv < 3) {
        uint32_t *cnts = &wave_cnt[wv][0][0];
        const uint32_t c = cnts[lane];
        const uint32_t incl = wave_incl_sum(c);
        cnts[lane] = incl - c;
        const uint32_t tot = __shfl(incl, kWave - 1, kWave);
        if (lane == 0) base[wv] = tot;
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
    for (int k = 0; k < kMgItems; ++k) {
        const int e = k * kMgThreads + t;
        if (f[k] & 2) {
            const int64_t p = (int64_t)base[0] + wave_cnt[0][k][wv] + mask_rank_lt(bt[k]);
            t_rank[p] = (uint64_t)slot_rank[e];
            t_suf[p] = slot_suf[e];
        }
        if (own_large[k]) {
            const int64_t p = (int64_t)base[1] + wave_cnt[1][k][wv] + mask_rank_lt(bl[k]);
            // Every group on the radix list has more than kG members, so the ranks of two of them differ by more than
            // kG >= 2^l_shift: rank >> l_shift still tells them apart, in order.  The sort field shrinks by l_shift bits
            // (one or two digit passes fewer); the low rank bits travel below it, where no pass looks.
            const uint64_t rr = (uint64_t)r[k];
            l_key[p] = l_shift > 0 ? (((rr >> l_shift) << (kbits + l_shift)) | ((uint64_t)k2[k] << l_shift) | (rr & ((1ull << l_shift) - 1)))
                                   : ((rr << kbits) | (uint64_t)k2[k]);
            l_suf[p] = s[k];
        }
        if (f[k] & 4) {
            const int64_t p = (int64_t)base[2] + wave_cnt[2][k][wv] + mask_rank_lt(bu[k]);
            if (u_ib > 0) {
                u_rank_end[-1 - p] = ((uint64_t)slot_rank[e] << u_ib) | (uint64_t)(ElemT)slot_suf[e];
            } else {
                u_rank_end[-1 - p] = (uint64_t)slot_rank[e];
                u_suf_end[-1 - p] = slot_suf[e];
            }
        }
    }
}

}  // namespace dq
