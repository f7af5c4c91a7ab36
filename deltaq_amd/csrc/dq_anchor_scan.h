// dq_anchor_scan.h -- step 1 of Diff.Create's scan loop (Diff.cs:100-125: Search at every position from the end of the
// last match, the oldscore bookkeeping, the break test) on the device, in ONE persistent launch per new file.
//
// The host loop of dq_bsdiff.h::scan_loop asks the device for a window of Search answers and waits: one dependent
// round trip (~30 us) per window, ~1 per edit between similar files -- 90 of the 100 ms of a 16 MiB pair.  Here the
// loop itself runs on the device and hands the host only what steps 2 and 3 (extensions, emission:
// dq_bsdiff.h::TripleEmitter) need: per control triple the position the loop broke on and where its match lies in
// old.  tests/anchor_model.py is the CPU model of exactly this evaluation, checked against a literal transcription of
// the reference's loop (tests/test_models_cpu.py).
//
// With agree(k) = "the previous alignment still gets byte k right" (k + shift < n and old[k + shift] == new[k]) and
// M_j = max(base, max_{k <= j}(k + len_k)), the loop's oldscore at the break test of position j is
//        carried_j = #{ k in [j, M_j) : agree(k) } = cnt(base, M_j) - cnt(base, j),
// so a whole WINDOW of positions is tested at once from two running sums (C = cnt(base, M), S = cnt(base, j)), a
// prefix count of agree() over the bytes the window covers and a prefix maximum of the match ends.  A position whose
// answer is not exact (its search hit the cap) or whose match reaches beyond the covered bytes is a STOP POINT: the
// window counts up to it, then it is taken on its own (exact search, a count over its whole match).
//
// Grid: kAsGroups workgroups of 4 waves, all resident.  A window's searches are dealt out one position per wave
// (128 positions; 65-ary search, dq_match_search.h) or, while the loop walks through a long differing stretch, one per
// lane (8192); the answers meet in device memory behind ONE grid barrier per window, and every workgroup then evaluates
// the window on its own -- the same integer decisions everywhere, so no second exchange is needed.  The barrier is an
// agent-scope counter (bounded spin, error flag); answers are agent-scope atomic stores / loads (the L2s of the 8 XCDs
// are not coherent with each other).
#pragma once
#include "dq_match_search.h"

namespace dq {

constexpr int kAsThreads = 256;
constexpr int kAsWaves = kAsThreads / kWave;
constexpr int kAsGroups = 32;
constexpr int kAsWaveWin = kAsGroups * kAsWaves;          // positions of a one-wave-per-position window
constexpr int kAsLaneWin = kAsWaveWin * kWave;            // ... of a one-lane-per-position window
constexpr int kAsExtra = 128;                             // bytes covered behind a window's last position
constexpr int64_t kAsCap = 64;                            // comparison cap of the speculative positions

struct AnchorCtl {
    unsigned long long arrive;                            // grid barrier: arrivals so far
    unsigned long long nrec;                              // (cursor, hit_pos) pairs written
    long long cursor, hit_len, hit_pos, shift;            // the loop's state at an anchor boundary (in and out)
    long long done;                                       // 1: the end of new has been reached and reported
    unsigned long long searches, windows, stops;          // Search calls the reference's loop makes; windows; stop points
    unsigned int error;                                   // 1: the barrier timed out
    unsigned int pad;
};

__device__ __forceinline__ int64_t as_wave_incl_max(int64_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const int64_t t = __shfl_up(v, o, kWave);
        if (l >= o) v = t > v ? t : v;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T as_wave_min(T v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const T t = __shfl_xor(v, o, kWave);
        v = t < v ? t : v;
    }
    return v;
}

template <typename IdxT>
__global__ __launch_bounds__(kAsThreads) void anchor_scan_kernel(
    const uint8_t *__restrict__ old, int64_t n, const IdxT *__restrict__ sa, const uint8_t *__restrict__ nw, int64_t m,
    const IdxT *__restrict__ ptab, int pk, unsigned long long *__restrict__ ans /* [kAsLaneWin]: len << 32 | pos */,
    int64_t *__restrict__ rec /* [rec_cap][2] */, int64_t rec_cap, AnchorCtl *__restrict__ ctl)
{
    __shared__ uint16_t agp[kAsLaneWin + kAsExtra + 2];   // agp[x] = #agree in [i, i + x)
    __shared__ int64_t w_i64[kAsWaves];
    __shared__ uint32_t w_u32[kAsWaves];
    __shared__ int32_t w_brk[kAsWaves], w_stp[kAsWaves];
    __shared__ int64_t s_pos, s_len;
    __shared__ uint32_t s_err;

    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wv = tid >> 6;
    const int gwave = blockIdx.x * kAsWaves + wv;         // this wave's place in the grid
    unsigned long long bar_target = 0;

    // ---- uniform state (every thread of every workgroup carries the same values) ----
    int64_t cursor = ctl->cursor, hit_len = ctl->hit_len, hit_pos = ctl->hit_pos, shift = ctl->shift;
    int64_t nrec = 0;
    unsigned long long n_search = 0, n_win = 0, n_stop = 0;
    bool failed = false;

    auto agree = [&](int64_t k) -> bool { return k + shift < n && old[k + shift] == nw[k]; };

    // #agree in [a, b), by the whole workgroup (every byte of a long match is counted once per anchor search)
    auto count_agree = [&](int64_t a, int64_t b) -> int64_t {
        const int64_t upto = b < n - shift ? b : n - shift;
        uint32_t c = 0;
        if (upto > a) {
            const uint8_t *po = old + shift, *pn = nw;
            constexpr uint64_t k7f = 0x7f7f7f7f7f7f7f7full;
            int64_t k = a + 8 * (int64_t)tid;
            // 8 bytes a step while 12 bytes exist in both buffers behind k (ms_load8 touches whole dwords)
            for (; k + 12 <= upto && k + shift + 12 <= n && k + 12 <= m; k += 8 * kAsThreads) {
                const uint64_t x = ms_load8(po + k) ^ ms_load8(pn + k);
                const uint64_t t = ~(((x & k7f) + k7f) | x | k7f);            // 0x80 in every byte of x that is zero
                c += (uint32_t)__popcll(t);
            }
            // the words this thread did not take that way (the tail of the range): byte by byte
            for (; k < upto; k += 8 * kAsThreads)
                for (int64_t q = k; q < k + 8 && q < upto; ++q) c += po[q] == pn[q];
        }
        c = wave_incl_sum(c);
        __syncthreads();
        if (lane == kWave - 1) w_u32[wv] = c;
        __syncthreads();
        int64_t tot = 0;
#pragma unroll
        for (int q = 0; q < kAsWaves; ++q) tot += w_u32[q];
        return tot;
    };

    auto grid_barrier = [&]() {
        __threadfence();
        __syncthreads();
        bar_target += gridDim.x;
        if (tid == 0) {
            __hip_atomic_fetch_add(&ctl->arrive, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t spins = 0, bad = 0;
            while (__hip_atomic_load(&ctl->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < bar_target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 26) || __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bad = 1;
                    break;
                }
            }
            s_err = bad;
        }
        __syncthreads();
        if (s_err) failed = true;
    };

    while (cursor < m && !failed) {
        if (nrec >= rec_cap) break;                       // the host empties the list and launches again from this state
        cursor += hit_len;
        const int64_t base = cursor;
        int64_t i = base, M = base, C = 0, S = 0;         // C = cnt(base, M), S = cnt(base, i), M >= i
        bool found = false, lane_mode = false;
        int streak = 0;                                   // stop points in a row that did not break
        int64_t carried_at = 0;
        bool any_search = false;
        int64_t last_pos = 0, last_len = 0;               // answer at the last position walked over
        while (i < m && !found && !failed) {
            const int64_t c = (m - i) < (lane_mode ? kAsLaneWin : kAsWaveWin) ? (m - i) : (lane_mode ? kAsLaneWin : kAsWaveWin);
            ++n_win;
            // ---- 0. the window's answers, one position per wave or per lane ----
            if (!lane_mode) {
                if (gwave < c) {
                    const int64_t scan = i + gwave;
                    const bool exact = gwave == 0 || streak >= 2;
                    int64_t p = 0, l = 0;
                    ms_search_wave<IdxT>(old, n, sa, nw, m, scan, exact ? (int64_t)0 : kAsCap, ptab, pk, &p, &l, nullptr, /*resume_first=*/true);
                    if (lane == 0)
                        __hip_atomic_store(&ans[gwave], ((unsigned long long)(uint32_t)(int32_t)l << 32) | (uint32_t)(int32_t)p,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                const int64_t idx = (int64_t)gwave * kWave + lane;
                if ((int64_t)gwave * kWave < c) {          // (whole waves)
                    int64_t p = 0, l = 0;
                    ms_search_one<IdxT>(old, n, sa, nw, m, idx < c ? i + idx : 0, idx < c, kAsCap, ptab, pk, &p, &l);
                    if (idx < c)
                        __hip_atomic_store(&ans[idx], ((unsigned long long)(uint32_t)(int32_t)l << 32) | (uint32_t)(int32_t)p,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            grid_barrier();
            if (failed) break;

            // ---- 1. prefix counts of agree() over the covered bytes [i, cover] ----
            const int64_t cover = (i + c + kAsExtra) < m ? (i + c + kAsExtra) : m;
            const int lc = (int)(cover - i);               // agp[0 .. lc]
            {
                const int seg = (lc + kAsThreads - 1) / kAsThreads;
                const int k0 = tid * seg < lc ? tid * seg : lc, k1 = (tid + 1) * seg < lc ? (tid + 1) * seg : lc;
                uint32_t mine = 0;
                for (int k = k0; k < k1; ++k) mine += agree(i + k) ? 1u : 0u;
                const uint32_t incl = wave_incl_sum(mine);
                if (lane == kWave - 1) w_u32[wv] = incl;
                __syncthreads();
                uint32_t run = incl - mine;
                for (int q = 0; q < wv; ++q) run += w_u32[q];
                for (int k = k0; k < k1; ++k) { agp[k] = (uint16_t)run; run += agree(i + k) ? 1u : 0u; }
                if (tid == 0) {                            // the total closes the array
                    uint32_t tot = 0;
                    for (int q = 0; q < kAsWaves; ++q) tot += w_u32[q];
                    agp[lc] = (uint16_t)tot;
                }
                __syncthreads();
            }

            // ---- 2. the positions, 256 at a time: stop rule, prefix maximum of the match ends, break test ----
            int64_t Mrun = M;                              // M after the positions walked over so far in this window
            int brk = -1, stp = -1;                        // window index of the first break / stop point
            const int64_t Cbase = C, Mbase = M;
            auto C_of = [&](int64_t Mj) -> int64_t {       // cnt(base, Mj) for Mj inside the coverage (or Mj == Mbase)
                return Mj > Mbase ? Cbase + (int64_t)agp[Mj - i] - (int64_t)agp[Mbase - i] : Cbase;
            };
            for (int ch = 0; ch * kAsThreads < c && brk < 0 && stp < 0; ++ch) {
                const int t = ch * kAsThreads + tid;
                const bool have = t < c;
                int64_t l = 0, e = -1;
                bool stop = false;
                if (have) {
                    const unsigned long long v = __hip_atomic_load(&ans[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    l = (int64_t)(int32_t)(uint32_t)(v >> 32);
                    const int64_t j = i + t;
                    if (l < 0) stop = true;
                    else {
                        e = j + l;
                        if (e > Mbase && e > cover) stop = true;
                    }
                    if (stop) e = -1;
                }
                // inclusive prefix maximum of e over the chunk, carried in from Mrun
                int64_t pm = as_wave_incl_max(e);
                if (lane == kWave - 1) w_i64[wv] = pm;
                __syncthreads();
                int64_t carry = Mrun;
                for (int q = 0; q < wv; ++q) carry = w_i64[q] > carry ? w_i64[q] : carry;
                int64_t chunk_max = Mrun;
                for (int q = 0; q < kAsWaves; ++q) chunk_max = w_i64[q] > chunk_max ? w_i64[q] : chunk_max;
                pm = pm > carry ? pm : carry;              // M_j
                bool brk_here = false;
                if (have && !stop) {
                    const int64_t carried = C_of(pm) - (S + (int64_t)agp[t]);
                    brk_here = (l == carried && l != 0) || l > carried + 8;
                }
                int32_t fb = brk_here ? t : 0x7fffffff, fs = (have && stop) ? t : 0x7fffffff;
                fb = as_wave_min(fb);
                fs = as_wave_min(fs);
                if (lane == 0) { w_brk[wv] = fb; w_stp[wv] = fs; }
                __syncthreads();
                int32_t b = 0x7fffffff, s2 = 0x7fffffff;
                for (int q = 0; q < kAsWaves; ++q) { b = w_brk[q] < b ? w_brk[q] : b; s2 = w_stp[q] < s2 ? w_stp[q] : s2; }
                __syncthreads();                           // (w_* are reused by the next chunk)
                if (b < s2) brk = b;
                else if (s2 != 0x7fffffff) stp = s2;
                if (brk < 0 && stp < 0) Mrun = chunk_max;
                else {
                    // M after the positions BEFORE the break / stop point: the prefix maximum just in front of it
                    const int at = brk >= 0 ? brk : stp;
                    const int owner = at - ch * kAsThreads;          // thread of this chunk that holds it
                    // (its own e is not part of what lies before it: take the exclusive value)
                    int64_t excl = as_wave_incl_max(e);
                    excl = __shfl_up(excl, 1, kWave);
                    if (lane == 0) excl = -1;
                    excl = excl > carry ? excl : carry;
                    if (tid == owner) s_pos = excl;
                    __syncthreads();
                    Mrun = s_pos;
                    __syncthreads();
                }
            }

            if (brk >= 0) {
                const unsigned long long v = __hip_atomic_load(&ans[brk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int64_t l = (int64_t)(int32_t)(uint32_t)(v >> 32), p = (int64_t)(int32_t)(uint32_t)v;
                const int64_t j = i + brk;
                int64_t Mj = Mrun;
                if (j + l > Mj) Mj = j + l;
                carried_at = C_of(Mj) - (S + (int64_t)agp[brk]);
                n_search += (unsigned long long)(j - base + 1);
                cursor = j; hit_pos = p; hit_len = l;
                found = true;
                break;
            }
            if (stp < 0) {                                 // the whole window went by
                const unsigned long long v = __hip_atomic_load(&ans[c - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_len = (int64_t)(int32_t)(uint32_t)(v >> 32);
                last_pos = (int64_t)(int32_t)(uint32_t)v;
                any_search = true;
                C = C_of(Mrun);
                M = Mrun;
                S += (int64_t)agp[c];
                i += c;
                lane_mode = true;                          // a long differing stretch: one position per lane from here on
            } else {
                // ---- 3. the stop point, on its own: exact answer, a count over its whole match ----
                ++n_stop;
                const int64_t j = i + stp;
                const unsigned long long v = __hip_atomic_load(&ans[stp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int64_t l = (int64_t)(int32_t)(uint32_t)(v >> 32), p = (int64_t)(int32_t)(uint32_t)v;
                if (l < 0) {
                    if (wv == 0) {
                        int64_t p2 = 0, l2 = 0;
                        ms_search_wave<IdxT>(old, n, sa, nw, m, j, 0, ptab, pk, &p2, &l2);
                        if (lane == 0) { s_pos = p2; s_len = l2; }
                    }
                    __syncthreads();
                    p = s_pos; l = s_len;
                    __syncthreads();
                }
                C = C_of(Mrun);
                M = Mrun;
                if (j + l > M) {
                    C += count_agree(M, j + l);
                    M = j + l;
                }
                const int64_t Sj = S + (int64_t)agp[stp];
                const int64_t carried = C - Sj;
                any_search = true;
                last_pos = p; last_len = l;
                if ((l == carried && l != 0) || l > carried + 8) {
                    n_search += (unsigned long long)(j - base + 1);
                    cursor = j; hit_pos = p; hit_len = l; carried_at = carried;
                    found = true;
                    break;
                }
                ++streak;
                S = Sj + (agree(j) ? 1 : 0);
                i = j + 1;
                lane_mode = false;
            }
            if (M < i) {                                   // (M >= i - 1 always: the last position walked over ends at or behind itself)
                C += agree(M) ? 1 : 0;
                M = i;
            }
        }
        if (failed) break;
        bool emit = true;
        if (!found) {
            n_search += (unsigned long long)(m - base > 0 ? m - base : 0);
            cursor = m;
            if (any_search) { hit_pos = last_pos; hit_len = last_len; }
        } else if (hit_len == carried_at) {
            emit = false;                                  // the old alignment explains it: keep scanning behind it
        }
        if (emit) {
            if (blockIdx.x == 0 && tid == 0) { rec[2 * nrec] = cursor; rec[2 * nrec + 1] = hit_pos; }
            ++nrec;
            shift = hit_pos - cursor;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        ctl->nrec = (unsigned long long)nrec;
        ctl->cursor = cursor; ctl->hit_len = hit_len; ctl->hit_pos = hit_pos; ctl->shift = shift;
        ctl->done = (!failed && cursor >= m) ? 1 : 0;
        ctl->searches = n_search; ctl->windows = n_win; ctl->stops = n_stop;
    }
}

}  // namespace dq
