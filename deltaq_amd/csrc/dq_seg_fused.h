// dq_seg_fused.h -- single-pass "rank rebucketing": group heads, device-wide
// (max, max, sum) scan, SA / ISA scatter and compaction of the still-tied suffixes in ONE
// read of the sorted list (the three-kernel seg_reduce / seg_scan / seg_apply pipeline of
// dq_sa_kernels.h read it twice and scanned the partials in a single workgroup).
//
// The device-wide scan is a decoupled look-back over per-tile status words, one 64-bit word
// per field {last new head, last parent-group head, active count}: [state:2 | value:62],
// each written by ONE agent-scope relaxed atomic store and polled with agent-scope relaxed
// atomic loads -- the word is its own flag, so the three fields need no mutual ordering and
// no fences (same protocol as radix_rank_kernel).  A whole wave inspects 64 predecessor tiles
// per step.  Tiles are handed out by an atomic ticket; spins are bounded.
//
// Inside a tile every wave owns 1024 consecutive list entries in wave-striped order
// (item k of lane l = entry 64 k + l: fully coalesced loads) and the scans are done on
// BALLOT MASKS instead of shuffles: H_k = ballot(head) gives "last head <= me" as a
// find-highest-bit below the lane and "actives before me" as a popcount below the lane, with
// wave-uniform carries from item to item.
#pragma once
#include "dq_sa_kernels.h"

namespace dq {

struct SegCtl {
    uint32_t ticket;
    uint32_t error;
};

constexpr uint64_t kSegAgg = 1ull << 62, kSegPrefix = 2ull << 62, kSegMask = (1ull << 62) - 1;
constexpr int kSegK = 16;                                   // entries per lane
constexpr int kSegWaveN = kWave * kSegK;                    // 1024 entries per wave
// 16 waves x 1024 entries = 16384 entries per tile.  Smaller tiles are capped by the ticket
// counter: a single device-scope counter hands out only ~88 tickets/us (MI355X_MICROARCH.md,
// row "dequeue"), i.e. >= 190 us for a 64 Mi-entry list cut into 4096-entry tiles.
// (Measured alternatives at 64 Mi entries: 256 threads x 16: 470 us; 1024 x 8 at 2
// workgroups/CU: 410-450 us; 1024 x 16: 390 us.  The kernel is instruction-bound: ~80 VALU
// per entry-item of 64-bit mask arithmetic.)
constexpr int kSegThreads = 1024;
constexpr int kSegWaves = kSegThreads / kWave;
constexpr int kSegFusedTile = kSegWaves * kSegWaveN;

// Exclusive prefix of field `st` for `tile`, combined with max (kSum = false) or + (kSum = true).
// Values are stored biased so that 0 is the identity of both.  Runs on one full wave.
template <bool kSum>
__device__ __forceinline__ uint64_t seg_lookback(uint64_t *st, int64_t tile, SegCtl *ctl,
                                                 int64_t *sticky_error, uint32_t spin_limit)
{
    const int lane = lane_id();
    uint64_t acc = 0;
    int64_t t0 = tile - 1;
    uint32_t spins = 0;
    for (;;) {
        const int64_t t = t0 - lane;
        const uint64_t s = t >= 0 ? __hip_atomic_load(st + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : kSegPrefix;                         // virtual empty prefix before tile 0
        const uint64_t state = s >> 62;
        uint64_t v = s & kSegMask;
        const uint64_t pre = __ballot(state == 2);
        const uint64_t notready = __ballot(state == 0);
        const int p = pre ? __builtin_ctzll(pre) : kWave;              // nearest tile with a prefix
        const uint64_t below = p >= kWave ? ~0ull : ((1ull << p) - 1);
        if (notready & below) {                                         // a nearer tile has not published yet
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) { atomicExch(&ctl->error, 1u); *sticky_error = 1; return acc; }
            continue;
        }
        if (lane > p) v = 0;
        const uint64_t red = kSum ? wave_sum(v) : wave_max(v);
        acc = kSum ? acc + red : (red > acc ? red : acc);
        if (p < kWave) return acc;
        t0 -= kWave;
    }
}

// kInitial: one parent group [0, m) of rank 0, keys = round-0 keys (>> kshift for packed words);
//           SA already holds the suffixes; ISA is written for every entry only with kWriteISA
//           (dense path predicted), otherwise it is built later if the dense path is taken.
// else:     composite keys (rank << kbits | key2); writes SA (kWriteSA) / ISA (kWriteISA).
// kEmitPairs (with kInitial): instead of the ISA scatter, one word per entry
//           (tied? << 63 | rank << kbits | suffix, kbits = bits of n-1) goes to act_rank in list order; the tied
//           suffixes are also appended, members of a group adjacent, to (list_rank, act_suf) with 32-bit ranks
//           (the 64-bit buffers are all busy until the words have been binned) when list_rank is given.
// totals[0] receives the number of still-active suffixes.
// rank_from_isa (doubling rounds only): the rank part of the composite keys is rank >> 1 -- still unique
// per group and order preserving, because groups in the tied list have >= 2 members -- and the
// true parent rank of an entry is read from ISA[suffix].
template <typename IdxT, bool kInitial, bool kWriteSA, bool kWriteISA, bool kEmitPairs = false>
__global__ __launch_bounds__(kSegThreads) void seg_fused_kernel(
    const uint64_t *__restrict__ keys, const IdxT *__restrict__ vals, int64_t m, int kbits, int kshift,
    IdxT *__restrict__ SA, IdxT *__restrict__ ISA, uint64_t *__restrict__ act_rank,
    IdxT *__restrict__ act_suf, uint64_t *__restrict__ status /*[3][ntiles]*/, int64_t ntiles,
    SegCtl *__restrict__ ctl, int64_t *__restrict__ totals, int64_t *__restrict__ sticky_error,
    int rank_from_isa = 0, uint32_t *__restrict__ list_rank = nullptr,
    int rank_lo = 0 /* > 0: the rank field holds rank >> rank_lo, the low rank_lo bits of a key (below kshift) the rest */,
    uint32_t spin_limit = kSpinLimit /* empty polls of the look-back before it gives up (DQ_FAULT=spin: 0) */)
{
    __shared__ int64_t w_nh[kSegWaves], w_gh[kSegWaves], w_cnt[kSegWaves];
    __shared__ uint64_t s_prefix[3];
    __shared__ uint32_t s_tile;

    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int w = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(&ctl->ticket, 1u);
    __syncthreads();
    const int64_t tile = s_tile;
    const int64_t wb = tile * kSegFusedTile + (int64_t)w * kSegWaveN;     // this wave's first entry
    // everything inside the wave's range is addressed with 32-bit offsets from wb
    const uint64_t *kp = keys + wb;
    const IdxT *vp = vals + wb;
    const int64_t left = m - wb;
    const int nv = left >= kSegWaveN ? kSegWaveN : (left > 0 ? (int)left : 0);   // entries that exist
    const IdxT wb_i = (IdxT)wb;

    // ---- coalesced, wave-striped load ----
    uint64_t ck[kSegK];
#pragma unroll
    for (int k = 0; k < kSegK; ++k) {
        const int e = k * kWave + lane;
        ck[k] = e < nv ? kp[e] >> kshift : 0;
    }
    // one entry of halo on each side of the wave's range (wave-uniform values)
    const uint64_t halo_prev = (wb > 0 && left >= 0) ? keys[wb - 1] >> kshift : 0;
    const bool has_after = left > kSegWaveN;
    const uint64_t halo_next = has_after ? keys[wb + kSegWaveN] >> kshift : 0;

    // ---- per item: ballot masks of group heads (H), parent-group heads (G) and of the members
    //      of groups of size > 1 (A).  A_k needs H_{k+1}, so it is finished one item later.  The
    //      masks are parked in LDS (wave-uniform 64-bit values x 48 would overflow the SGPR file) ----
    __shared__ uint64_t m_H[kSegWaves][kSegK], m_G[kSegWaves][kSegK], m_A[kSegWaves][kSegK];
    int wave_nh = -1, wave_gh = -1, wave_cnt = 0;                     // local (offset from wb), or -1
    uint64_t Hprev = 0, Vprev = 0;                                   // masks of item k-1
#pragma unroll
    for (int k = 0; k <= kSegK; ++k) {
        uint64_t Hk = 0, Gk = 0, Vk = 0;
        if (k < kSegK) {
            const int e = k * kWave + lane;
            // key of the previous entry: lane-1 of this item (DPP wave shift), lane 63 of the
            // previous item, or the halo
            const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)ck[k], 0x138, 0xf, 0xf, false);
            const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(ck[k] >> 32), 0x138, 0xf, 0xf, false);
            uint64_t pk = ((uint64_t)phi << 32) | plo;
            uint64_t edge = halo_prev;
            if (k > 0) {
                const uint64_t q = ck[k > 0 ? k - 1 : 0];
                edge = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(q >> 32), kWave - 1) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)q, kWave - 1);
            }
            if (lane == 0) pk = edge;
            const bool valid = e < nv;
            const bool first = (wb == 0 && e == 0);
            Hk = __ballot(valid && (first || ck[k] != pk));
            Gk = kInitial ? (wb == 0 && k == 0 ? 1ull : 0ull)
                          : __ballot(valid && (first || (ck[k] >> kbits) != (pk >> kbits)));
            const int rem = nv - k * kWave;
            Vk = rem >= kWave ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1));
            if (Hk) wave_nh = k * kWave + (63 - __builtin_clzll(Hk));
            if (Gk) wave_gh = k * kWave + (63 - __builtin_clzll(Gk));
            if (lane == 0) { m_H[w][k] = Hk; if (!kInitial) m_G[w][k] = Gk; }
        }
        if (k > 0) {
            // finish item k-1: an entry is active unless it is a head AND the next entry is a head
            // (entries past the end of the list count as heads)
            const uint64_t hx = Hprev | ~Vprev;
            uint64_t next0;
            if (k < kSegK) next0 = (Hk | ~Vk) & 1ull;
            else {
                const uint64_t q = ck[kSegK - 1];
                const uint64_t last_key =
                    ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(q >> 32), kWave - 1) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)q, kWave - 1);
                next0 = (!has_after || halo_next != last_key) ? 1ull : 0ull;
            }
            const uint64_t Ak = Vprev & ~(Hprev & ((hx >> 1) | (next0 << 63)));
            wave_cnt += __popcll(Ak);
            if (lane == 0) m_A[w][k - 1] = Ak;
        }
        Hprev = Hk; Vprev = Vk;
    }
    if (lane == 0) {
        w_nh[w] = wave_nh >= 0 ? wb + wave_nh : -1;
        w_gh[w] = wave_gh >= 0 ? wb + wave_gh : -1;
        w_cnt[w] = wave_cnt;
    }
    __syncthreads();

    // ---- publish the tile's aggregates and look back: wave f handles field f ----
    if (w < 3) {
        uint64_t agg = 0;
        if (w == 2) { for (int i = 0; i < kSegWaves; ++i) agg += (uint64_t)w_cnt[i]; }
        else {
            int64_t mx = -1;
            for (int i = 0; i < kSegWaves; ++i) { const int64_t v = (w == 0 ? w_nh[i] : w_gh[i]); mx = v > mx ? v : mx; }
            agg = (uint64_t)(mx + 1);
        }
        uint64_t *st = status + (int64_t)w * ntiles;
        if (lane == 0)
            __hip_atomic_store(st + tile, (tile == 0 ? kSegPrefix : kSegAgg) | agg, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint64_t pf = 0;
        if (tile > 0) {
            pf = (w == 2) ? seg_lookback<true>(st, tile, ctl, sticky_error, spin_limit)
                          : seg_lookback<false>(st, tile, ctl, sticky_error, spin_limit);
            const uint64_t incl = (w == 2) ? pf + agg : (agg > pf ? agg : pf);
            if (lane == 0)
                __hip_atomic_store(st + tile, kSegPrefix | incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_prefix[w] = pf;
        if (w == 2 && lane == 0 && tile == ntiles - 1) totals[0] = (int64_t)(pf + agg);
    }
    __syncthreads();

    // ---- running values at the start of this wave: tile prefix + earlier waves of the tile ----
    int64_t cn64 = (int64_t)s_prefix[0] - 1, cg64 = (int64_t)s_prefix[1] - 1, cc = (int64_t)s_prefix[2];
    for (int i = 0; i < w; ++i) {
        cn64 = w_nh[i] > cn64 ? w_nh[i] : cn64;
        cg64 = w_gh[i] > cg64 ? w_gh[i] : cg64;
        cc += w_cnt[i];
    }
    IdxT cn = (IdxT)cn64, cg = (IdxT)cg64;           // list positions fit the index type

    // ---- apply ----
    const uint64_t le = (2ull << lane) - 1;          // bits at or below my lane
    const uint64_t lt = le >> 1;                     // bits strictly below
    const uint64_t lb = 1ull << lane;
    // issue every suffix load first (independent, in flight together)
    IdxT suf[kSegK];
#pragma unroll
    for (int k = 0; k < kSegK; ++k) {
        const int e = k * kWave + lane;
        const bool need = (kWriteSA || kWriteISA || kEmitPairs) ? (e < nv) : ((m_A[w][k] & lb) != 0);
        suf[k] = need ? vp[e] : (IdxT)0;
    }
#pragma unroll
    for (int k = 0; k < kSegK; ++k) {
        const int e = k * kWave + lane;
        const uint64_t Hk = m_H[w][k], Ak = m_A[w][k];
        const uint64_t Gk = kInitial ? (wb == 0 && k == 0 ? 1ull : 0ull) : m_G[w][k];
        if ((kWriteSA || kWriteISA || kEmitPairs) ? (k * kWave < nv) : (Ak != 0)) {          // wave-uniform skip
            const bool act = (Ak & lb) != 0;
            if ((kWriteSA || kWriteISA || kEmitPairs) ? (e < nv) : act) {
                const uint64_t hm = Hk & le, gm = Gk & le;
                const IdxT rn = hm ? wb_i + (IdxT)(k * kWave + 63 - __builtin_clzll(hm)) : cn;
                const IdxT rg = gm ? wb_i + (IdxT)(k * kWave + 63 - __builtin_clzll(gm)) : cg;
                // rank_from_isa: the composite key carries rank >> 1 (n close to 2^32, dq_sorter_impl.h::run);
                // a tied suffix's own ISA entry is its parent rank
                IdxT rank = kInitial ? (IdxT)0 : (kWriteISA && rank_from_isa) ? ISA[suf[k]] : (IdxT)(ck[k] >> kbits);
                // (the radix list of an LDS-class round: rank >> rank_lo above the key, the low bits as payload below it)
                if (!kInitial && rank_lo > 0) rank = (IdxT)(((uint64_t)rank << rank_lo) | (kp[e] & ((1ull << rank_lo) - 1)));
                const IdxT nr = rank + (rn - rg);
                if (kEmitPairs) {
                    // (tied?, rank, suffix) in list order, coalesced: the inverse suffix array is built from
                    // these words after they have been binned by suffix (dq_isa_pairs.h); kbits = bits of n-1
                    act_rank[wb + e] = ((uint64_t)(act ? 1 : 0) << 63) | ((uint64_t)nr << kbits) | (uint64_t)suf[k];
                    if (list_rank && act) {
                        const int64_t o = cc + __popcll(Ak & lt);
                        list_rank[o] = (uint32_t)nr;
                        act_suf[o] = suf[k];
                    }
                } else {
                    // dense doubling (ISA maintained) never reads SA again: a still-tied member's slot is
                    // written once, in the round that resolves it
                    if (kWriteSA && !(kWriteISA && act)) SA[rank + (wb_i + (IdxT)e - rg)] = suf[k];
                    // ISA[s] already holds the parent rank: only members whose rank moved need a (random) write
                    if (kWriteISA && (kInitial || nr != rank)) ISA[suf[k]] = nr;
                    if (act) {
                        const int64_t o = cc + __popcll(Ak & lt);
                        act_rank[o] = (uint64_t)nr;
                        act_suf[o] = suf[k];
                    }
                }
            }
        }
        if (Hk) cn = wb_i + (IdxT)(k * kWave + 63 - __builtin_clzll(Hk));
        if (Gk) cg = wb_i + (IdxT)(k * kWave + 63 - __builtin_clzll(Gk));
        cc += __popcll(Ak);
    }
}

}  // namespace dq
