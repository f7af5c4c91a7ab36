// dq_ties.h -- tie groups of a packed round-0 sort without re-reading the sorted keys.
//
// The last digit pass (radix_rank_kernel<kKeysLastTies>) already holds every tile in sorted order,
// so it records "key[o] == key[o-1]" as one bit per output position (ebits) for all pairs inside a
// tile's digit run.  What it cannot see are the pairs whose two members come from different tiles:
// the last key of one tile's run for digit d and the first key of the next tile's run for d are
// neighbours in the output.  The pass leaves those two words per (tile, digit) in seam_tab:
//
//   tie_seam_kernel     one thread per (tile, digit) run: compares its first key with the last key
//                       of the nearest earlier tile that has the digit; equal -> sets the bit of the
//                       run's first output position (known from the pass's final status words)
//   tie_collect_kernel  one thread per 64 positions of ebits: emits (rank = position of the group's
//                       first member, suffix = SA[p]) for every member of a group of size > 1.
//                       A group is emitted by the thread that owns its first member, so members stay
//                       adjacent; lists are appended with one atomic per workgroup (order across
//                       workgroups is arbitrary, as in dq_small_groups.h).
//
// This replaces a full pass over the sorted words (8 B per suffix) by 1 bit per suffix.  Runs of more
// than kTieMaxRun equal keys (long repeats in an otherwise random text) set *overflow and the host
// falls back to the general rebucket pass.
#pragma once
#include "dq_onesweep.h"

namespace dq {

constexpr int kTieMaxRun = 4096;
constexpr int kTieThreads = 1024;

template <typename StatusT>
__global__ __launch_bounds__(kBlock) void tie_seam_kernel(const uint64_t *__restrict__ seam_tab, int64_t ntiles,
                                                          int ib, const int64_t *__restrict__ digit_offset,
                                                          const StatusT *__restrict__ status,
                                                          uint32_t *__restrict__ ebits)
{
    using SB = StatusBits<StatusT>;
    const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= ntiles * kRadixSize) return;
    const int64_t t = idx / kRadixSize;
    const int d = (int)(idx % kRadixSize);
    const uint64_t first = seam_tab[idx * 2];
    if (first == kSeamEmpty || t == 0) return;
    int64_t tp = t - 1;
    while (tp >= 0 && seam_tab[(tp * kRadixSize + d) * 2] == kSeamEmpty) --tp;
    if (tp < 0) return;
    const uint64_t last = seam_tab[(tp * kRadixSize + d) * 2 + 1];
    if ((last >> ib) != (first >> ib)) return;
    // keys with digit d in tiles 0..t-1 = the inclusive prefix tile t-1 left in its status word
    const int64_t o = digit_offset[d] + (int64_t)(status[(t - 1) * kRadixSize + d] & SB::kMask);
    atomicOr(&ebits[(uint64_t)o >> 5], 1u << ((uint32_t)o & 31u));
}

struct TieCounters {
    unsigned long long count;       // entries appended
    unsigned long long overflow;    // a run longer than kTieMaxRun was met
};

template <typename IdxT>
__global__ __launch_bounds__(kTieThreads) void tie_collect_kernel(const uint64_t *__restrict__ ebits, int64_t nwords,
                                                                  int64_t n, const IdxT *__restrict__ SA,
                                                                  uint64_t *__restrict__ act_rank,
                                                                  IdxT *__restrict__ act_suf,
                                                                  TieCounters *__restrict__ ctr)
{
    __shared__ uint32_t wave_tot[kTieThreads / kWave];
    __shared__ unsigned long long s_base;
    const int lane = lane_id();
    const int w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kTieThreads + threadIdx.x;

    // E bit b: position 64*i + b ties with its predecessor.  A member of a group of size > 1 has its
    // own bit or its successor's bit set.
    uint64_t E = 0, own = 0;
    uint32_t tail = 0;                                   // members beyond this word (the run goes on)
    bool over = false;
    if (i < nwords) {
        E = ebits[i];
        const uint64_t next0 = (i + 1 < nwords) ? (ebits[i + 1] & 1ull) : 0ull;
        const uint64_t A = E | (E >> 1) | (next0 << 63);
        // leading positions that continue a group started in an earlier word belong to that word's thread
        const int lead = (E & 1ull) ? (E == ~0ull ? 64 : __builtin_ctzll(~E)) : 0;
        own = lead >= 64 ? 0ull : (A & (~0ull << lead));
        if (own >> 63 && next0) {
            // my last group runs on into the following words
            int64_t j = i + 1;
            for (;;) {
                const uint64_t F = j < nwords ? ebits[j] : 0ull;
                const int c = F == ~0ull ? 64 : __builtin_ctzll(~F);
                tail += (uint32_t)c;
                if (c < 64) break;
                if (tail > (uint32_t)kTieMaxRun) { over = true; break; }
                ++j;
            }
        }
    }
    if (over) { atomicExch(&ctr->overflow, 1ull); tail = 0; own = 0; }
    const uint32_t cnt = (uint32_t)__popcll(own) + tail;
    const uint32_t incl = wave_incl_sum(cnt);
    if (lane == kWave - 1) wave_tot[w] = incl;
    __syncthreads();
    uint32_t off = incl - cnt, tot = 0;
#pragma unroll
    for (int k = 0; k < kTieThreads / kWave; ++k) {
        const uint32_t c = wave_tot[k];
        if (k < w) off += c;
        tot += c;
    }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(&ctr->count, (unsigned long long)tot) : 0ull;
    __syncthreads();
    // ---- emission.  Many ties: the whole wave on one word at a time, lane b takes bit b, so the SA reads and
    //      the list writes of a word are coalesced (a lane walking its own 64 bits wrote 16 bytes at a time: 16.6 ms
    //      for the 4.75e8 tied suffixes of a 2 GiB random text, against 5.0 ms this way) ----
    const int64_t out = (int64_t)s_base + off;
    // (few ties -- the usual case after a 32...36-bit key: most words have no member at all and a lane walking its
    // own few bits is cheaper than the wave visiting every non-empty word)
    const uint32_t wave_members = __shfl(incl, kWave - 1, kWave);
    if (wave_members < 512u) {
        int64_t o = out, head = 0;
        uint64_t rest = own;
        while (rest) {
            const int b = __builtin_ctzll(rest);
            rest &= rest - 1;
            const int64_t p = i * 64 + b;
            if (!((E >> b) & 1ull)) head = p;               // first member of a group
            act_rank[o] = (uint64_t)head;
            act_suf[o] = SA[p];
            ++o;
        }
        for (uint32_t k = 0; k < tail; ++k) {
            act_rank[o] = (uint64_t)head;
            act_suf[o] = SA[(i + 1) * 64 + k];
            ++o;
        }
        return;
    }
    uint64_t todo = __ballot(cnt != 0);
    const uint64_t lbit = 1ull << lane;
    while (todo) {
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        const uint64_t own_j = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(own >> 32), j) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)own, j);
        const uint64_t E_j = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(E >> 32), j) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)E, j);
        const int64_t out_j = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)out >> 32), j) << 32) |
                                        (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)out, j));
        const uint32_t tail_j = (uint32_t)__builtin_amdgcn_readlane((int)tail, j);
        const int64_t p0 = (i - lane + j) * 64;                         // first position of word j of this wave
        if (own_j & lbit) {
            const uint64_t below = lbit - 1;
            const int64_t p = p0 + lane;
            // head of my group: me, or the nearest position below me that does not tie with its predecessor
            // (it exists inside the word: positions continuing an earlier word's group are not in own)
            const int64_t head = (E_j & lbit) ? p0 + (63 - __builtin_clzll(~E_j & below)) : p;
            const int64_t o = out_j + __popcll(own_j & below);
            act_rank[o] = (uint64_t)head;
            act_suf[o] = SA[p];
        }
        if (tail_j) {
            // my last group runs on into the following words: its head is the highest non-tie bit of the word
            const int64_t head = p0 + (63 - __builtin_clzll(~E_j));
            const int64_t o0 = out_j + __popcll(own_j);
            for (uint32_t k = (uint32_t)lane; k < tail_j; k += kWave) {
                act_rank[o0 + k] = (uint64_t)head;
                act_suf[o0 + k] = SA[p0 + 64 + k];
            }
        }
    }
    (void)n;
}

}  // namespace dq
