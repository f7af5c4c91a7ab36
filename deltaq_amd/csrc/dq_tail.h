// dq_tail.h -- the last doubling rounds of a sort in ONE launch: once at most kTailMax suffixes are still tied, one
// workgroup keeps the list in LDS and runs round after round until nothing is tied.
//
// A short list is pure latency in the device-wide rounds: mid_group_round_kernel + isa_update_kernel are two launches
// of ~20 + ~7 us whatever the list's length (one tile, one workgroup), a chain of 8 rounds ends in a host round trip, and
// a long repeat keeps a few hundred suffixes tied for log2(repeat length / h) rounds.  Nothing in a round needs more
// than one workgroup at that size -- and with ONE workgroup the reason for the deferred rank updates is gone too (no
// other workgroup gathers while this one writes), so a round is: gather, place, write, compact, next.
//
//   gather    key2 = ISA[s + h] + h | n - 1 - s past the end (the shorter suffix first); a member inside a run takes the
//             rank behind its run (dq_runs.h) -- exactly gather_key2_kernel's rule.  The loads are agent-scope (they
//             bypass this CU's L1, which the workgroup's own rank stores of the round before do not refresh).
//   place     a member's slot inside its group = #(smaller key2) + #(equal key2 before it): one walk over the group's
//             key2 values in LDS, as mid_group_round_kernel does -- for groups of ANY size up to the whole list
//   write     alone with its key2 -> SA[rank + smaller] = s; smaller > 0 -> ISA[s] = rank + smaller
//   compact   the members that still share (rank, key2) stay, in slot order: groups adjacent, ranks ascending
//
// 64 KiB of text: the last rounds of the sort (a list of ~3000 entries) in one launch instead of 6 x 2; a 199 000-byte
// repeat in 1.5 GiB of random bytes: 15 rounds over 2 x 199 000 ... 2 entries.
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kTailThreads = 1024;
constexpr int kTailItems = 4;
constexpr int kTailMax = kTailThreads * kTailItems;       // 4096 list entries
constexpr int kTailWaves = kTailThreads / kWave;
constexpr int kTailMaxRounds = 80;                        // h at least doubles: 64 rounds exhaust any int64 length

struct TailResult {
    unsigned long long rounds;        // rounds run
    unsigned long long entries;       // list entries summed over the rounds (dq_last_sort_info's sum_active)
    unsigned long long left;          // entries still tied at the end (0 unless the round bound was hit)
};

// KeyT / kSteps: key2 values as 32-bit words and three of them per member -- (ISA[s + h], ISA[s + 2h], ISA[s + 3h]), the depth
// grows 4-fold a round, see dq_mid_groups.h -- where rank + h fits 32 bits (int32 indices) and no run lengths are in force;
// one 64-bit key otherwise.
template <typename IdxT, typename KeyT = uint64_t, int kSteps = 1>
__global__ __launch_bounds__(kTailThreads) void tail_rounds_kernel(
    const uint64_t *__restrict__ rank_in, const IdxT *__restrict__ suf_in, int m, int64_t n, int64_t h,
    IdxT *__restrict__ ISA, IdxT *__restrict__ SA, const uint32_t *__restrict__ RL /* run lengths, or none */,
    TailResult *__restrict__ res,
    const unsigned long long *__restrict__ m_dev = nullptr /* launched behind a chain of rounds, speculatively: the list
                                                             length is what the last of them left (low 32 bits); a list
                                                             still longer than kTailMax is left alone (res->left says so) */)
{
    // (n < 2^32 on this path: ranks and suffixes fit 32 bits; key2 = rank + h may not)
    __shared__ uint32_t s_rank[2][kTailMax], s_suf[2][kTailMax];
    __shared__ KeyT s_key[kSteps][kTailMax];
    __shared__ int16_t s_g0[kTailMax];                     // first list position of the entry's group
    __shared__ uint16_t s_gsz[kTailMax];                   // by head position: members of the group
    __shared__ uint8_t s_flag[kTailMax];                   // by slot: 1 resolved, 2 still tied, 4 rank moved
    __shared__ int s_wmax[kTailWaves];
    __shared__ uint32_t s_wcnt[kTailWaves];
    __shared__ int s_m;

    const int t = threadIdx.x;
    const int lane = lane_id();
    const int wv = t >> 6;
    int cur = 0;
    if (m_dev) {
        const unsigned long long real = *m_dev & 0xffffffffull;
        if (real > (unsigned long long)kTailMax) {
            if (t == 0) { res->rounds = 0; res->entries = 0; res->left = real; }
            return;
        }
        m = (int)real;
    }
    for (int i = t; i < m; i += kTailThreads) {
        s_rank[0][i] = (uint32_t)rank_in[i];
        s_suf[0][i] = (uint32_t)suf_in[i];
    }
    __syncthreads();

    unsigned long long rounds = 0, entries = 0;
    while (m > 0 && rounds < (unsigned long long)kTailMaxRounds) {
        ++rounds;
        entries += (unsigned long long)m;
        // ---- gather (striped: i = k * kTailThreads + t) ----
#pragma unroll
        for (int k = 0; k < kTailItems; ++k) {
            const int i = k * kTailThreads + t;
            if (i < m) {
                const int64_t s = (int64_t)s_suf[cur][i];
                int64_t off = h;
                if (RL) {
                    const uint32_t r = RL[s];
                    if ((int64_t)r > h) off = (int64_t)r;
                }
                const int64_t q = s + off;
                uint64_t k2;
                if (q < n) k2 = (uint64_t)__hip_atomic_load(&ISA[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (uint64_t)h;
                else k2 = off > h ? 0ull : (uint64_t)(n - 1 - s);
                s_key[0][i] = (KeyT)k2;
#pragma unroll
                for (int j = 1; j < kSteps; ++j) {                // (kSteps > 1: no run lengths, off == h)
                    const int64_t qj = s + (int64_t)(j + 1) * h, qp = s + (int64_t)j * h;
                    uint64_t kj;
                    if (qj < n) kj = (uint64_t)__hip_atomic_load(&ISA[qj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (uint64_t)h;
                    else kj = qp < n ? (uint64_t)(n - 1 - s - (int64_t)j * h) : 0ull;
                    s_key[j][i] = (KeyT)kj;
                }
            }
        }
        // ---- group starts (blocked: thread t owns positions 4 t .. 4 t + 3): max-scan of the head positions ----
        {
            const int p0 = t * kTailItems;
            int run = -1, hp[kTailItems];
#pragma unroll
            for (int k = 0; k < kTailItems; ++k) {
                const int p = p0 + k;
                const bool head = p < m && (p == 0 || s_rank[cur][p] != s_rank[cur][p - 1]);
                run = head ? p : run;
                hp[k] = run;
            }
            const int wincl = wave_incl_max(run);
            if (lane == kWave - 1) s_wmax[wv] = wincl;
            int carry = __shfl_up(wincl, 1, kWave);
            if (lane == 0) carry = -1;
            __syncthreads();
            for (int i = 0; i < wv; ++i) carry = s_wmax[i] > carry ? s_wmax[i] : carry;
#pragma unroll
            for (int k = 0; k < kTailItems; ++k) {
                const int p = p0 + k;
                if (p < m) s_g0[p] = (int16_t)(hp[k] > carry ? hp[k] : carry);
            }
        }
        __syncthreads();
        // group sizes, by head: every head closes the group before it, the end of the list the last one (a counted
        // walk below: its loads do not wait for each other, as they would behind a "same group?" test per step)
#pragma unroll
        for (int k = 0; k < kTailItems; ++k) {
            const int p = k * kTailThreads + t;
            if (p < m) {
                const int g = s_g0[p];
                if (g == p && p > 0) { const int pg = s_g0[p - 1]; s_gsz[pg] = (uint16_t)(p - pg); }
                if (p == m - 1) s_gsz[g] = (uint16_t)(m - g);
            }
        }
        __syncthreads();
        // ---- place: walk the group ----
        int slot[kTailItems];
        uint32_t nrank[kTailItems], mysuf[kTailItems];
        uint8_t flag[kTailItems];
#pragma unroll
        for (int k = 0; k < kTailItems; ++k) {
            const int i = k * kTailThreads + t;
            slot[k] = -1; nrank[k] = 0; mysuf[k] = 0; flag[k] = 0;
            if (i < m) {
                const int g0 = s_g0[i];
                KeyT mine[kSteps];
#pragma unroll
                for (int q = 0; q < kSteps; ++q) mine[q] = s_key[q][i];
                int less = 0, eq = 0, eq_before = 0;
                const int gs = s_gsz[g0], me = i - g0;
                for (int j = 0; j < gs; ++j) {
                    const KeyT o = s_key[0][g0 + j];
                    bool lt = o < mine[0], same = o == mine[0];
#pragma unroll
                    for (int q = 1; q < kSteps; ++q) {
                        const KeyT oq = s_key[q][g0 + j];
                        lt = lt || (same && oq < mine[q]);
                        same = same && oq == mine[q];
                    }
                    less += lt;
                    eq += same;
                    eq_before += same && j < me;
                }
                slot[k] = g0 + less + eq_before;
                nrank[k] = s_rank[cur][i] + (uint32_t)less;
                mysuf[k] = s_suf[cur][i];
                flag[k] = (uint8_t)((eq > 1 ? 2 : 1) | (less != 0 ? 4 : 0));
            }
        }
        __syncthreads();                                  // every reader of s_rank / s_suf / s_key / s_g0 of this round is done
        const int nxt = cur ^ 1;
#pragma unroll
        for (int k = 0; k < kTailItems; ++k) {
            if (slot[k] >= 0) {
                s_rank[nxt][slot[k]] = nrank[k];
                s_suf[nxt][slot[k]] = mysuf[k];
                s_flag[slot[k]] = flag[k];
            }
        }
        __syncthreads();
        // ---- write what is decided, keep what is not (blocked, so that the kept entries stay in slot order) ----
        {
            const int p0 = t * kTailItems;
            uint32_t keep[kTailItems], kr[kTailItems], ks[kTailItems];
            uint32_t cnt = 0;
#pragma unroll
            for (int k = 0; k < kTailItems; ++k) {
                const int p = p0 + k;
                keep[k] = 0; kr[k] = 0; ks[k] = 0;
                if (p < m) {
                    const uint8_t f = s_flag[p];
                    const uint32_t r = s_rank[nxt][p], s = s_suf[nxt][p];
                    if (f & 1) SA[r] = (IdxT)s;            // alone with its key: rank + #smaller is its place
                    if (f & 4) __hip_atomic_store(&ISA[s], (IdxT)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (f & 2) { keep[k] = 1; kr[k] = r; ks[k] = s; ++cnt; }
                }
            }
            const uint32_t incl = wave_incl_sum(cnt);
            if (lane == kWave - 1) s_wcnt[wv] = incl;
            __syncthreads();                              // (also: every read of buffer nxt above is done before it is overwritten)
            uint32_t base = incl - cnt;
            for (int i = 0; i < wv; ++i) base += s_wcnt[i];
            if (t == kTailThreads - 1) s_m = (int)(base + cnt);
            // compact in place into buffer nxt: position base + (kept before me) <= p always, and the barrier above
            // separated every read from these writes
#pragma unroll
            for (int k = 0; k < kTailItems; ++k) {
                if (keep[k]) { s_rank[nxt][base] = kr[k]; s_suf[nxt][base] = ks[k]; ++base; }
            }
        }
        // The rank stores of this round and the next round's gathers are agent-scope atomics of waves of ONE workgroup:
        // the barrier orders them (release / acquire at workgroup scope is all two waves of a workgroup need), and
        // neither goes through this CU's L1.  (An agent-scope fence pair here -- a write-back of the whole L2's dirty
        // lines per round -- cost more than the rounds: 5 rounds of a 3600-entry list 135 us with)
        __syncthreads();
        m = s_m;
        cur = nxt;
        h *= (kSteps + 1);
        __syncthreads();                                  // (s_m is rewritten next round)
    }
    if (t == 0) { res->rounds = rounds; res->entries = entries; res->left = (unsigned long long)m; }
}

}  // namespace dq
