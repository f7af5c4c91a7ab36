// dq_anchor_scan.h -- step 1 of Diff.Create's scan loop (Diff.cs:100-125: Search at every position from the end of the
// last match, the oldscore bookkeeping, the break test) on the device, in ONE persistent launch per new file.
//
// The host loop of dq_bsdiff.h::scan_loop asks the device for a window of Search answers and waits: one dependent
// round trip (~30 us) per window, ~1.5 per edit between similar files -- 90 of the 100 ms of a 16 MiB pair.  Here the
// loop itself runs on the device and hands the host only what steps 2 and 3 (extensions, emission:
// dq_bsdiff.h::TripleEmitter) need: per control triple the position the loop broke on and where its match lies in
// old.  tests/anchor_model.py is the CPU model of this evaluation, checked against a literal transcription of the
// reference's loop (tests/test_models_cpu.py).
//
// With agree(k) = "the previous alignment still gets byte k right" (k + shift < n and old[k + shift] == new[k]),
// cnt(a, b) = #{ k in [a, b) : agree(k) } and M_j = max(base, max_{k <= j}(k + len_k)), the loop's oldscore at the
// break test of position j is
//        carried_j = cnt(j, M_j) = cnt(base, M_j) - cnt(base, j).
// Every searching wave (lane) also counts cw_j = cnt(j, j + len_j) over its own match, so that
// cnt(base, j + len_j) = S_j + cw_j with S_j = cnt(base, j) -- a prefix count over the window's POSITIONS only.  A
// whole window is then tested at once: prefix maximum of the match ends j + len_j carrying cnt(base, .) of the end
// that holds the maximum, minus S_j.  A position whose answer is not exact (its search hit the cap; or the count of its
// long match is an upper bound that does not settle its break test, see wave_cw) is a STOP POINT: the window counts up
// to it, then it is searched again exactly and taken on its own.
//
// Grid: G = kAsGroups workgroups of 4 waves, all resident.  A window's searches are dealt out one position per wave
// (4 G = 512 positions; 65-ary search, dq_match_search.h) or, while the loop walks through a long differing stretch, one
// per lane (256 G = 32768).  The answers meet in device memory; every workgroup evaluates the window on its own -- the
// same integer decisions everywhere, so nothing but the answers is exchanged -- the first 64 positions, then 256 at a
// time, as their answers turn up: every answer word carries a tag of its window, so an evaluator waits on the very words
// it needs and on nothing else (no grid-wide barrier, no arrival flags; a lagging barrier of one window keeps the two
// answer buffers apart).  Every exchanged word is an agent-scope atomic (the L2s of the 8 XCDs are not coherent with
// each other); every spin is bounded and ends in an error flag, on which the host falls back to its own loop.
//
// Several grids on one file (dq_diff.hip, "chains").  What the loop does from the end of an iteration on is a function of
// two numbers only -- where the iteration ended and the shift in force (an emitted triple sets the shift to hit_pos -
// cursor, and Search(cursor) is a function of cursor) -- so a second grid started speculatively in the middle of the file,
// under a shift no real alignment can have, walks the same iterations as the grid that comes from the front as soon as
// the two end ONE iteration at the same place under the same shift; between similar files that is the first or second
// iteration.  Every iteration end is therefore written to the list (emitted ones as before, the others with bit 63 set),
// a grid can be told to leave after `extra` iteration ends behind `stop_at` (the start of the next grid) or, in the
// middle of an iteration, after `lane_budget` one-lane-per-position windows (unrelated data: speculation buys nothing
// there), and a grid can resume in the middle of an iteration from the state another launch left.
#pragma once
#include "dq_match_search.h"

namespace dq {

constexpr int kAsThreads = 256;
constexpr int kAsWaves = kAsThreads / kWave;
constexpr int kAsGroups = 128;                            // workgroups of the persistent grid (DQ_SCAN_GROUPS: 8 .. kAsMaxGroups;
                                                          // Diff.Create of 16 MiB random / text pairs with 2000 edits:
                                                          // 64 groups 59 / 94 ms, 96: 57 / 92, 128: 55 / 92)
constexpr int kAsMaxGroups = 128;
constexpr int kAsMaxLaneWin = kAsMaxGroups * kAsWaves * kWave;   // positions of the largest one-lane-per-position window
constexpr int64_t kAsCap = 64;                            // comparison cap of the speculative positions
constexpr int kAsWaveWins = 3;                            // wave windows walked over before the lane windows take over: those of
constexpr int kAsWaveWinsMax = 16;                        // a whole grid of kAsGroups workgroups (1536 positions); a narrower grid
                                                          // walks as many positions in more of them, 16 at most (a window costs
                                                          // ~15 us whatever its width, a lane window ~0.4 ms: 8 grids of 32 workgroups
                                                          // on the 16 MiB pair with 2000 edits of up to 400 bytes 46.9 ms before, 4 x 64 20.0)
// An answer is two words, each with the tag of its window in the top two bits: tag << 62 | len << 31 | pos, and
// tag << 62 | cnt(j, j + len).  The windows of a kind (one position per wave / per lane) take the two answer buffers of
// their kind in turn; the k-th use of a buffer has tag k % 3, the host fills the buffers with ones (tag 3) before a
// launch.  Every use of a buffer writes ALL the slots a later use can read (a window is shorter than its kind's size
// only at the end of the file), so what a reader can find in a slot is this use's word or the use before's -- never a
// word of three uses ago with the same tag.  (With one pair of buffers for both kinds, a slot beyond the 512 of a wave
// window kept its word from the last lane window however long ago: tests/manual/stress_bsdiff.py seed 421.)
constexpr unsigned long long kAsStopLen = 0x7fffffffull;  // len field of a stop point (texts stay below 2^31 - 1 bytes)
constexpr int kAsFirstSpan = 64;                          // positions of a wave window evaluated in the first step
constexpr int64_t kAsCountFront = 4096;                   // bytes of a long match under another alignment that are counted (one step)
constexpr int64_t kAsFront = 512;                         // bytes behind every window position fetched ahead of its search
constexpr unsigned long long kAsBoundBit = 1ull << 61;    // second answer word: the count is an upper bound

struct AnchorCtl {
    unsigned long long nrec;                              // (cursor, hit_pos) pairs written
    long long cursor, hit_len, hit_pos, shift;            // the loop's state at an anchor boundary (in and out)
    long long done;                                       // 1: the end of new has been reached and reported
    unsigned long long searches, windows, stops;          // Search calls the reference's loop makes; windows; stop points
    unsigned int error;                                   // 1: the barrier timed out
    unsigned int pad;                                     // in: bit 0 = fill in the times below; bits 8..13 = s_sleep of a poller;
                                                          // bits 16..20 = log2 of the spin bound (0: 24); bits 24..31: test knob,
                                                          // workgroup (k - 1) sleeps before it publishes a window's answers
    unsigned long long t_search, t_wait, t_eval, t_stop;  // workgroup 0's time searching / waiting for answers / evaluating /
                                                          // at stop points, in 100 MHz ticks (DQ_TRACE prints them)
    // ---- several grids on one file ----
    long long stop_at, extra;                             // in: leave after `extra` iteration ends at cursor >= stop_at (extra <= 0: never)
    long long lane_budget;                                // in: leave, in the middle of an iteration, before one-lane-per-position window
                                                          // number lane_budget + 1 of this launch (0: never)
    long long mid;                                        // in: 1 = resume in the middle of an iteration from the fields below;
                                                          // out: 1 = left in the middle of one (they are filled in)
    long long base, i, M, C, S, last_pos, last_len;
    int lane_mode, streak, passed, any_search;
    unsigned long long t_begin, t_end;                    // out: the device's 100 MHz clock when workgroup 0 began and ended (DQ_TRACE)
};
constexpr unsigned long long kAsSilent = 1ull << 63;      // list entry of an iteration end that emitted nothing: kAsSilent | cursor << 32
static_assert(sizeof(AnchorCtl) <= 248, "a control block and the word behind it fill 256 bytes");

// One launch carries up to kScanMaxChains grids ("chains"): workgroup b belongs to chain b / groups and works on that
// chain's slot of the buffers below.  (One launch on one stream, not a launch per chain on streams of their own: streams
// share the device's few hardware queues, and a chain's completion was reported one whole kernel late behind another
// chain's -- 16 MiB pair with 2000 edits 26 ms, 21 ms with GPU_MAX_HW_QUEUES=8.  For the same reason a chain's result does
// not wait for the stream either: the chain writes it to pinned memory itself, the launch's number behind it.)
constexpr int kScanMaxChains = 16;
constexpr int64_t kAnchorRecs = 1 << 15;                  // list entries per chain and launch
constexpr size_t kAnchorAnswers = ((size_t)kAsMaxLaneWin + (size_t)kAsMaxGroups * kAsWaves) * 32;    // two buffers of either kind, 16 B a slot
// device scratch: control blocks (in: the state to start from; the error word), completion words, answer buffers
constexpr size_t kAsFinishedAt = (size_t)kScanMaxChains * 256;
constexpr size_t kAsAnswersAt = kAsFinishedAt + (size_t)kScanMaxChains * 2048;
constexpr size_t kAnchorScratch = kAsAnswersAt + (size_t)kScanMaxChains * kAnchorAnswers;
static_assert(kAsMaxGroups * 8 <= 2048, "completion words");
// pinned host memory: upload blocks, then per slot the list, the Search counts beside it, the result block and, in its
// last word, the number of the launch that wrote it
constexpr size_t kAsSlotAt = (size_t)kScanMaxChains * 256;
constexpr size_t kAsSlotBytes = (size_t)kAnchorRecs * 16 + 256;
constexpr size_t kAnchorPinned = kAsSlotAt + (size_t)kScanMaxChains * kAsSlotBytes;
struct AnchorLaunch {
    int chains, groups;                                   // grids in this launch, workgroups of each
    int slot[kScanMaxChains];                             // buffers of chain k
    unsigned long long seq;                               // this launch's number (never 0)
};

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

// #{ k in [a, b) : po[k] == pn[k] } by one wave (wave-uniform arguments): 4 KiB a step, then 512 bytes a step, while 12
// more bytes exist in both buffers behind every lane's 8 (ms_load8 touches whole dwords), then 64 bytes a step.  lim_o / lim_n: bytes that
// exist in the two buffers counted from index 0 of po / pn.
__device__ __forceinline__ int64_t as_wave_count_equal(const uint8_t *po, int64_t lim_o, const uint8_t *pn, int64_t lim_n,
                                                       int64_t a, int64_t b)
{
    const int lane = lane_id();
    constexpr uint64_t k7f = 0x7f7f7f7f7f7f7f7full;
    uint32_t c = 0;
    int64_t k = a;
    while (k + 8 * 512 + 4 <= b && k + 8 * 512 + 4 <= lim_o && k + 8 * 512 + 4 <= lim_n) {      // 4 KiB a step, all in flight
        uint64_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t q = k + u * 512 + 8 * lane;
            x[u] = ms_load8(po + q) ^ ms_load8(pn + q);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) c += (uint32_t)__popcll(~(((x[u] & k7f) + k7f) | x[u] | k7f));   // 0x80 in every byte of x that is zero
        k += 8 * 512;
    }
    while (k + 512 + 4 <= b && k + 512 + 4 <= lim_o && k + 512 + 4 <= lim_n) {
        const int64_t q = k + 8 * lane;
        const uint64_t x = ms_load8(po + q) ^ ms_load8(pn + q);
        c += (uint32_t)__popcll(~(((x & k7f) + k7f) | x | k7f));
        k += 512;
    }
    for (; k < b; k += kWave) {
        const int64_t q = k + lane;
        c += (q < b && po[q] == pn[q]) ? 1u : 0u;
    }
    return (int64_t)__shfl(wave_incl_sum(c), kWave - 1, kWave);
}

inline size_t as_agp_bytes(int groups) { return ((size_t)groups * kAsThreads + 2) * 2 + 12 & ~(size_t)15; }

// kMinBlocks 1: all the registers the search wants (352), one workgroup per compute unit; 2: 256 registers, 117 spilled
// (a window ~11 % slower), two workgroups per compute unit -- for launches of more narrow grids than the device holds
// otherwise (16 MiB pair with 2000 edits, Diff.Create: 8 grids of 32 with all registers 14.0 ms, 16 with half 12.5;
// 20 000 small edits 40.7 / 31.7)
template <typename IdxT, int kMinBlocks>
__global__ __launch_bounds__(kAsThreads, kMinBlocks) void anchor_scan_kernel(
    const uint8_t *__restrict__ old, int64_t n, const IdxT *__restrict__ sa, const uint8_t *__restrict__ nw, int64_t m,
    const IdxT *__restrict__ ptab, int pk, char *__restrict__ scratch, char *__restrict__ pinned, const AnchorLaunch ln)
{
    const int n_groups = ln.groups;                       // workgroups of this chain
    const int chain = (int)blockIdx.x / n_groups, bid = (int)blockIdx.x - chain * n_groups;
    const int slot = ln.slot[chain];
    AnchorCtl *__restrict__ ctl = reinterpret_cast<AnchorCtl *>(scratch + (size_t)slot * 256);      // read-only here but for the error word
    // [n_groups], zeroed: windows workgroup w has finished evaluating
    unsigned long long *__restrict__ finished = reinterpret_cast<unsigned long long *>(scratch + kAsFinishedAt + (size_t)slot * 2048);
    // [2][lane window][2] then [2][wave window][2]: tagged answers, all ones before the launch
    unsigned long long *__restrict__ ans = reinterpret_cast<unsigned long long *>(scratch + kAsAnswersAt + (size_t)slot * kAnchorAnswers);
    // PINNED HOST memory: the list (cursor << 32 | hit_pos, one store each), Search calls of the reference's loop up to
    // each entry, and what this launch leaves (the host does not wait for the stream to learn it)
    unsigned long long *__restrict__ rec = reinterpret_cast<unsigned long long *>(pinned + kAsSlotAt + (size_t)slot * kAsSlotBytes);
    unsigned long long *__restrict__ cum = rec + kAnchorRecs;
    AnchorCtl *__restrict__ out = reinterpret_cast<AnchorCtl *>(rec + 2 * kAnchorRecs);
    constexpr int64_t rec_cap = kAnchorRecs;
    // agp[x] = cnt(i, i + x), x = 0 .. c: as_agp_bytes(groups) of dynamic LDS -- a narrow grid's workgroups are small, and
    // more of them fit the device at once (16 grids of 32 workgroups: 16 KB each; 64 KB for a grid of 128)
    extern __shared__ uint16_t agp[];
    const int64_t kAsWaveWin = (int64_t)n_groups * kAsWaves;       // positions of a one-wave-per-position window
    const int64_t kAsLaneWin = kAsWaveWin * kWave;                 // ... of a one-lane-per-position window
    const int wave_wins = (int)((kAsWaveWins * kAsGroups * kAsWaves + kAsWaveWin - 1) / kAsWaveWin) < kAsWaveWinsMax
                              ? (int)((kAsWaveWins * kAsGroups * kAsWaves + kAsWaveWin - 1) / kAsWaveWin) : kAsWaveWinsMax;
    __shared__ int64_t w_e[kAsWaves], w_c[kAsWaves];
    __shared__ uint32_t w_u32[kAsWaves];
    __shared__ int32_t w_brk[kAsWaves], w_stp[kAsWaves];
    __shared__ int64_t s_v[4];
    __shared__ uint32_t s_err;
    // one-lane-per-position windows: what every workgroup says about its own 256 positions (see "2L" below)
    __shared__ unsigned long long sm_key[kAsMaxGroups], sm_res[kAsMaxGroups][5];
    __shared__ uint32_t sm_agree[kAsMaxGroups];

    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wv = tid >> 6;
    const int gwave = bid * kAsWaves + wv;                // this wave's place in its chain's grid

    // ---- uniform state (every thread of every workgroup carries the same values) ----
    int64_t cursor = ctl->cursor, hit_len = ctl->hit_len, hit_pos = ctl->hit_pos, shift = ctl->shift;
    int64_t nrec = 0;
    unsigned long long n_search = ctl->searches, n_win = 0, n_stop = 0;     // (the Search count runs on from launch to launch)
    const int64_t stop_at = ctl->stop_at, extra = ctl->extra, lane_budget = ctl->lane_budget;
    int64_t past = 0;                                     // iteration ends at cursor >= stop_at so far
    bool resume = ctl->mid != 0, left_mid = false;
    int64_t mid_v[7] = {0, 0, 0, 0, 0, 0, 0};
    int mid_f[4] = {0, 0, 0, 0};
    unsigned long long n_wave_win = 0, n_lane_win = 0;  // windows of either kind so far
    bool failed = false;
    unsigned long long t_search = 0, t_wait = 0, t_eval = 0, t_stop = 0, t0 = __builtin_readcyclecounter();
    // (only when asked for -- ctl->pad, set under DQ_TRACE: reading the clock ~7 times a window is not free)
    const bool timed = (ctl->pad & 1u) != 0;
    const int poll_sleep = (int)((ctl->pad >> 8) & 63u);  // (from the host: DQ_SCAN_POLL_SLEEP, 16 by default)
    // every spin is bounded: 2^24 polls (seconds) in production; the fault-injection tests send a bound of a few polls
    // (DQ_FAULT=spin), under which the launch reports its error and the host loop takes the file
    const uint32_t spin_bound = 1u << (((ctl->pad >> 16) & 31u) ? ((ctl->pad >> 16) & 31u) : 24u);
    // (DQ_SCAN_SLOW_GROUP=k: workgroup k - 1 is the straggler of every window -- the adversarial schedule of the tests)
    const bool slow_me = ((ctl->pad >> 24) & 255u) == (unsigned)bid + 1u;
    auto lap = [&](unsigned long long &acc) {
        if (!timed) return;
        const unsigned long long t1 = wall_clock64();
        acc += t1 - t0;
        t0 = t1;
    };
    t0 = wall_clock64();
    const unsigned long long t_begin = t0;

    auto agree = [&](int64_t k) -> bool { return k + shift < n && old[k + shift] == nw[k]; };
    // cnt(j, j + l) for the match (p, l) found at position j, by one wave: a match under the previous alignment agrees
    // everywhere; otherwise the bytes are compared (the part of [j, j + l) that the shifted old file still covers)
    // A long match under another alignment is counted over its first kAsCountFront bytes only, the rest taken as agreeing:
    // an UPPER BOUND (*bound = true).  The evaluation lets such a position break where its length beats even the bound --
    // the usual case: a new alignment after an edit, the old one wrong nearly everywhere -- and makes it a stop point
    // (searched and counted again, exactly) where the bound would have to be carried on (tests/anchor_model.py, `loose`).
    // 80 kB between two edits: 20 dependent steps of 4 KiB less for the one wave the window waits for.
    auto wave_cw = [&](int64_t j, int64_t p, int64_t l, bool *bound) -> int64_t {
        if (bound) *bound = false;
        if (l <= 0) return 0;
        if (p - j == shift) return l;
        const int64_t upto = (j + l) < (n - shift) ? (j + l) : (n - shift);
        if (upto <= j) return 0;
        if (bound && upto - j > kAsCountFront + 512) {
            *bound = true;
            return as_wave_count_equal(old + shift, n - shift, nw, m, j, j + kAsCountFront) + (upto - j - kAsCountFront);
        }
        return as_wave_count_equal(old + shift, n - shift, nw, m, j, upto);
    };

    // No grid-wide barrier.  An evaluator loads the answer words of the positions it is about to look at and spins on
    // them until both carry this window's tag -- the loop usually breaks near the front of a window, the slowest of
    // its 512 searches rarely lies there.  Two answer buffers are taken in turn, so nobody may publish window W before
    // EVERYBODY is through with window W - 2: finished[w] counts the evaluations workgroup w has completed (a lagging
    // barrier: it holds only a workgroup that is two windows ahead; one word per workgroup, loaded before the search
    // and looked at after it).  Progress: the workgroups at the smallest window index never wait for anybody behind them.
    // Everything exchanged goes through agent-scope relaxed atomics (stores / loads that bypass the non-coherent cache
    // levels), as radix_rank_kernel's status words do: no fences, so the read-only data (old, new, suffix array,
    // prefix table) stays cached from window to window; and no ordering between words is relied upon -- each one
    // says itself which window it belongs to.
    auto give_up = [&]() {
        __hip_atomic_store(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_err = 1;
    };
    auto others_gave_up = [&]() -> bool { return __hip_atomic_load(&ctl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; };
    // (every workgroup on its own word -- a sum over the workgroups says nothing: half of them two windows ahead
    // would make it look as if everybody had finished one)
    auto lagging_load = [&](unsigned long long win_no) -> unsigned long long {       // issued before the search ...
        if (win_no < 2 || tid >= n_groups) return ~0ull;
        return __hip_atomic_load(&finished[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto await_lagging = [&](unsigned long long win_no, unsigned long long seen) {   // ... checked before publishing window win_no
        if (win_no >= 2 && tid < n_groups) {
            uint32_t spins = 0;
            while (seen < win_no - 1) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > spin_bound || ((spins & 1023u) == 0 && others_gave_up())) { give_up(); break; }
                seen = __hip_atomic_load(&finished[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (s_err) failed = true;
    };
    unsigned long long n_finished = 0;
    auto window_done = [&]() {
        ++n_finished;
        if (tid == 0) __hip_atomic_store(&finished[bid], n_finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // (two answer buffers, taken in turn by the windows: a workgroup that is through with window k publishes window
    // k + 1 while a slower one still reads window k; nobody reaches window k + 2 before everybody has left window k)
    unsigned long long *ans_w = ans;
    unsigned long long tag_w = 0;
    auto publish = [&](int64_t slot, int64_t p, int64_t l, int64_t cw, bool bound) {
        unsigned long long *ans = ans_w;
        const unsigned long long lf = l < 0 ? kAsStopLen : (unsigned long long)l, pf = l < 0 ? 0ull : (unsigned long long)p;
        __hip_atomic_store(&ans[2 * slot], (tag_w << 62) | (lf << 31) | pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ans[2 * slot + 1], (tag_w << 62) | (bound ? kAsBoundBit : 0ull) | (unsigned long long)cw, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    };
    // both words of position `slot` of this window, waited for
    auto fetch = [&](int64_t slot, unsigned long long *v, unsigned long long *v2) {
        unsigned long long *ans = ans_w;
        uint32_t spins = 0;
        for (;;) {
            *v = __hip_atomic_load(&ans[2 * slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *v2 = __hip_atomic_load(&ans[2 * slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((*v >> 62) == tag_w && (*v2 >> 62) == tag_w) return;
            // (thousands of lanes poll while the slowest searches still run, and uncached polls are fabric traffic: how
            // long a poller sleeps is what it costs the searches.  s_sleep 1 / 4 / 16 / 32: Diff.Create of the 16 MiB pairs
            // with 2000 and 20 000 edits 44.6 / 42.6 / 42.5 / 42.5 ms and 277 / 271 / 268 / 270 ms; the host sends 16)
            if (poll_sleep <= 1) __builtin_amdgcn_s_sleep(1);
            else if (poll_sleep == 2) __builtin_amdgcn_s_sleep(2);
            else if (poll_sleep <= 4) __builtin_amdgcn_s_sleep(4);
            else if (poll_sleep <= 8) __builtin_amdgcn_s_sleep(8);
            else if (poll_sleep <= 16) __builtin_amdgcn_s_sleep(16);
            else __builtin_amdgcn_s_sleep(32);
            if (++spins > spin_bound || ((spins & 1023u) == 0 && others_gave_up())) { give_up(); *v = *v2 = 0; return; }
        }
    };
    if (tid == 0) s_err = 0;                               // (sticky: once set, every loop below ends)
    __syncthreads();

    while (cursor < m && !failed) {
        if (nrec >= rec_cap) break;                       // the host empties the list and launches again from this state
        if (extra > 0 && past >= extra) break;            // (far enough behind the start of the next grid)
        if (!resume) cursor += hit_len;
        int64_t base = cursor;
        int64_t i = base, M = base, C = 0, S = 0;         // C = cnt(base, M), S = cnt(base, i), M >= i
        bool found = false, lane_mode = false;
        int streak = 0;                                   // stop points in a row that did not break
        int passed = 0;                                   // wave windows in a row that went by without a break
        int64_t carried_at = 0;
        bool any_search = false;
        int64_t last_pos = 0, last_len = 0;               // answer at the last position walked over
        if (resume) {                                     // the middle of an iteration, as another launch left it
            base = ctl->base; i = ctl->i; M = ctl->M; C = ctl->C; S = ctl->S; last_pos = ctl->last_pos; last_len = ctl->last_len;
            lane_mode = ctl->lane_mode != 0; streak = ctl->streak; passed = ctl->passed; any_search = ctl->any_search != 0;
            resume = false;
        }
        while (i < m && !found && !failed) {
            if (lane_budget > 0 && (int64_t)n_lane_win >= lane_budget) {       // a long differing stretch: leave between two windows
                left_mid = true;
                mid_v[0] = base; mid_v[1] = i; mid_v[2] = M; mid_v[3] = C; mid_v[4] = S; mid_v[5] = last_pos; mid_v[6] = last_len;
                mid_f[0] = lane_mode ? 1 : 0; mid_f[1] = streak; mid_f[2] = passed; mid_f[3] = any_search ? 1 : 0;
                break;
            }
            const int64_t win = lane_mode ? kAsLaneWin : kAsWaveWin;
            const int64_t c = (m - i) < win ? (m - i) : win;
            const unsigned long long win_no = n_win;        // windows are numbered from 0, the same everywhere
            {
                // (the buffers of this window's kind, by the count of windows of that kind)
                unsigned long long &uses = lane_mode ? n_lane_win : n_wave_win;
                unsigned long long *kind = lane_mode ? ans : ans + 4 * (size_t)kAsMaxLaneWin;
                const size_t slots = lane_mode ? (size_t)kAsMaxLaneWin : (size_t)kAsMaxGroups * kAsWaves;
                ans_w = kind + (size_t)(uses & 1ull) * (2 * slots);
                tag_w = (uses >> 1) % 3ull;
                ++uses;
            }
            ++n_win;
            const unsigned long long lag_seen = lagging_load(win_no);
            // ---- 0. the window's answers (and each match's own agree count), one position per wave or per lane ----
            int64_t ln_p = 0, ln_l = 0, ln_cw = 0;          // (a lane window's answers stay with the lanes that searched)
            bool ln_live = false;
            {
                int64_t p = 0, l = 0, cw = 0, slot = -1;
                bool cw_bound = false;
                if (!lane_mode) {
                    if (gwave < c) {
                        const int64_t scan = i + gwave;
                        const bool exact = gwave == 0 || streak >= 2;
                        // The first 512 bytes behind the position under the CURRENT alignment, 8 per lane, asked for before
                        // the search (the addresses are known: their latency hides behind the first probes): the agree
                        // count of a match that ends inside them is read off the lanes' words, and for a longer one they
                        // are the counted front of the upper bound -- no dependent step of its own in either case.
                        int64_t front = n - shift - scan < m - scan ? n - shift - scan : m - scan;       // bytes both texts have
                        front = front - 4 < kAsFront ? ((front - 4) & ~(int64_t)7) : kAsFront;           // (ms_load8 touches whole dwords)
                        uint64_t fx = ~0ull;
                        if (8 * lane + 8 <= front) fx = ms_load8(old + shift + scan + 8 * lane) ^ ms_load8(nw + scan + 8 * lane);
                        ms_search_wave<IdxT>(old, n, sa, nw, m, scan, exact ? (int64_t)0 : kAsCap, ptab, pk, &p, &l, nullptr, /*resume_first=*/true);
                        const int64_t len_here = (scan + l < n - shift ? scan + l : n - shift) - scan;   // bytes of the match agree() can hold for
                        if (l > 0 && p - scan != shift && len_here > 0 && front >= 64 && (len_here <= front || len_here > kAsFront + 64)) {
                            const int64_t take = len_here < front ? len_here : front;                    // counted bytes: [0, take)
                            const int64_t mine = take - 8 * lane;                                        // ... of this lane's 8
                            constexpr uint64_t k7f = 0x7f7f7f7f7f7f7f7full;
                            uint64_t eq = ~(((fx & k7f) + k7f) | fx | k7f);                              // 0x80 in every byte the two texts agree in
                            if (mine <= 0) eq = 0;
                            else if (mine < 8) eq &= (1ull << (8 * mine)) - 1;
                            const uint32_t c = (uint32_t)__popcll(eq);
                            cw = (int64_t)__shfl(wave_incl_sum(c), kWave - 1, kWave) + (len_here - take);  // (+ the bytes behind the front, taken as agreeing)
                            cw_bound = len_here > take;
                        } else {
                            cw = wave_cw(scan, p, l, &cw_bound);
                        }
                        if (lane == 0) slot = gwave;
                    }
                } else {
                    const int64_t idx = (int64_t)gwave * kWave + lane;
                    if ((int64_t)gwave * kWave < c) {      // (whole waves)
                        const bool live = idx < c;
                        ms_search_one<IdxT>(old, n, sa, nw, m, live ? i + idx : 0, live, kAsCap, ptab, pk, &p, &l);
                        if (live) {
                            const int64_t j = i + idx;
                            if (l > 0) {
                                if (p - j == shift) cw = l;
                                else for (int64_t k = j; k < j + l; ++k) cw += agree(k) ? 1 : 0;
                            }
                            ln_p = p; ln_l = l; ln_cw = cw; ln_live = true;
                        }
                    }
                }
                lap(t_search);
                await_lagging(win_no, lag_seen);           // (window win_no - 2 has been read by everybody)
                lap(t_wait);
                if (failed) break;
                if (slow_me) {                             // (test knob: ~50 us behind everybody else, every window)
                    for (int z = 0; z < 64; ++z) __builtin_amdgcn_s_sleep(127);
                }
                if (slot >= 0) publish(slot, p, l, cw, cw_bound);
            }

            int64_t Mrun = M, Crun = C;                    // (M, cnt(base, M)) after the positions walked over so far
            int brk = -1, stp = -1;                        // window index of the first break / stop point
            int64_t b_pos = 0, b_len = 0, b_carried = 0;
            int64_t win_agree = 0, stp_agree = 0;          // cnt(i, i + c); cnt(i, stop point)
            if (!lane_mode) {
            // ---- 1. prefix counts of agree() over the window's positions ----
            {
                const int lc = (int)c;
                const int seg = (lc + kAsThreads - 1) / kAsThreads;
                const int k0 = tid * seg < lc ? tid * seg : lc, k1 = (tid + 1) * seg < lc ? (tid + 1) * seg : lc;
                uint32_t bits[(kAsMaxLaneWin / kAsThreads + 31) / 32] = {0};
                uint32_t mine = 0;
                for (int k = k0; k < k1; ++k) {
                    const uint32_t a = agree(i + k) ? 1u : 0u;
                    bits[(k - k0) >> 5] |= a << ((k - k0) & 31);
                    mine += a;
                }
                const uint32_t incl = wave_incl_sum(mine);
                if (lane == kWave - 1) w_u32[wv] = incl;
                __syncthreads();
                uint32_t run = incl - mine;
                for (int q = 0; q < wv; ++q) run += w_u32[q];
                for (int k = k0; k < k1; ++k) { agp[k] = (uint16_t)run; run += (bits[(k - k0) >> 5] >> ((k - k0) & 31)) & 1u; }
                if (tid == 0) {                            // the total closes the array
                    uint32_t tot = 0;
                    for (int q = 0; q < kAsWaves; ++q) tot += w_u32[q];
                    agp[lc] = (uint16_t)tot;
                }
                __syncthreads();
            }

            // ---- 2. the positions, a step at a time: prefix maximum of the match ends (with cnt(base, end)), break test ----
            // (one-wave-per-position windows: the first 64 positions on their own -- a deletion or a copied stretch breaks
            // on the first few -- then 256 a step; one-lane-per-position windows: 256 positions a step, one workgroup's answers)
            for (int off = 0; off < (int)c && brk < 0 && stp < 0;) {
                const int span = (!lane_mode && off == 0) ? kAsFirstSpan : kAsThreads;
                lap(t_eval);
                const int t = off + tid;
                const bool have = tid < span && t < c;
                int64_t l = 0, p = 0, e = -1, ce = 0, Sj = 0;
                bool stop = false, bound = false;
                if (have) {
                    unsigned long long v, v2;
                    fetch(t, &v, &v2);
                    const unsigned long long lf = (v >> 31) & 0x7fffffffull;
                    p = (int64_t)(v & 0x7fffffffull);
                    Sj = S + (int64_t)agp[t];
                    if (lf == kAsStopLen) { stop = true; l = -1; }
                    else {
                        l = (int64_t)lf; e = i + t + l;
                        ce = Sj + (int64_t)(v2 & 0x7fffffffull);                  // cnt(base, e) = cnt(base, j) + cnt(j, e)
                        bound = (v2 & kAsBoundBit) != 0;
                    }
                }
                off += span;
                lap(t_wait);
                // inclusive prefix maximum of e over the chunk, carrying cnt(base, .) of the end that holds it: both in ONE
                // word per lane (end relative to the window + 1 above, the count's complement below -- equal ends carry
                // the same count, cnt(base, e) is a function of e, unless one of them is an upper bound: then the
                // smaller count is the exact one and the complement makes the maximum pick it), one shuffle a round
                const uint64_t key = e < 0 ? 0ull : (((uint64_t)(uint32_t)(e - i + 1)) << 32) | (uint64_t)(0xffffffffu - (uint32_t)ce);
                uint64_t pk = key;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const uint64_t tk = __shfl_up(pk, o, kWave);
                    if (lane >= o && tk > pk) pk = tk;
                }
                auto end_of = [&](uint64_t k) -> int64_t { return k ? (int64_t)(k >> 32) - 1 + i : (int64_t)-1; };
                auto cnt_of = [&](uint64_t k) -> int64_t { return k ? (int64_t)(0xffffffffu - (uint32_t)k) : (int64_t)0; };
                const int64_t pe = end_of(pk), pc = cnt_of(pk);
                if (lane == kWave - 1) { w_e[wv] = pe; w_c[wv] = pc; }
                __syncthreads();
                if (s_err) { failed = true; break; }       // (an answer never came: everybody leaves)
                int64_t ce_in = Mrun, cc_in = Crun;        // what comes in from the left of this wave
                for (int q = 0; q < wv; ++q) if (w_e[q] > ce_in) { ce_in = w_e[q]; cc_in = w_c[q]; }
                int64_t ch_e = Mrun, ch_c = Crun;          // ... and what the whole chunk leaves
                for (int q = 0; q < kAsWaves; ++q) if (w_e[q] > ch_e) { ch_e = w_e[q]; ch_c = w_c[q]; }
                // exclusive values (what the positions BEFORE this one left), then this position's own
                const uint64_t xk = __shfl_up(pk, 1, kWave);
                int64_t xe = end_of(xk), xc = cnt_of(xk);
                if (lane == 0 || xe <= ce_in) { xe = ce_in; xc = cc_in; }
                int64_t Mj = xe, Cj = xc;
                if (e > Mj) { Mj = e; Cj = ce; }
                bool brk_here = false;
                int64_t carried = 0;
                if (have && !stop) {
                    carried = Cj - Sj;
                    brk_here = (l == carried && l != 0) || l > carried + 8;
                    // its own end is the running maximum and its count a bound: carried is at most this -- a break
                    // that needs no more is one; anything else is decided with the exact count, on its own
                    if (bound && e > xe && !(l > carried + 8)) { brk_here = false; stop = true; }
                }
                // (the lanes of a wave hold consecutive positions: the first lane that says so is the first position)
                const uint64_t mb = __ballot(brk_here), ms = __ballot(have && stop);
                const int32_t t0w = off - span + (wv << 6);                       // position of this wave's lane 0
                const int32_t fb = mb ? t0w + __builtin_ctzll(mb) : 0x7fffffff, fs = ms ? t0w + __builtin_ctzll(ms) : 0x7fffffff;
                if (lane == 0) { w_brk[wv] = fb; w_stp[wv] = fs; }
                __syncthreads();
                int32_t b = 0x7fffffff, s2 = 0x7fffffff;
                for (int q = 0; q < kAsWaves; ++q) { b = w_brk[q] < b ? w_brk[q] : b; s2 = w_stp[q] < s2 ? w_stp[q] : s2; }
                if (b < s2) brk = b;
                else if (s2 != 0x7fffffff) stp = s2;
                if (brk < 0 && stp < 0) { Mrun = ch_e; Crun = ch_c; }
                else {
                    // the thread that holds the break / stop point hands over what lay before it (and, for a break, its answer)
                    const int at = brk >= 0 ? brk : stp;
                    if (t == at) { s_v[0] = xe; s_v[1] = xc; s_v[2] = (p << 32) | (l & 0xffffffffll); s_v[3] = carried; }
                    __syncthreads();
                    Mrun = s_v[0]; Crun = s_v[1];
                    b_pos = s_v[2] >> 32; b_len = (int64_t)(int32_t)(uint32_t)s_v[2]; b_carried = s_v[3];
                }
                __syncthreads();                           // (w_* and s_v are reused)
            }
            win_agree = (int64_t)agp[c];
            if (stp >= 0) stp_agree = (int64_t)agp[stp];
            if (!failed && brk < 0 && stp < 0) {
                // (the window's last position was in the last step: the word is this window's; never a stop point here)
                const unsigned long long v = __hip_atomic_load(&ans_w[2 * (c - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_len = (int64_t)((v >> 31) & 0x7fffffffull);
                last_pos = (int64_t)(v & 0x7fffffffull);
            }
            } else {
            // ---- 2L. one position per lane: every workgroup evaluates ITS OWN 256 positions, two small exchanges ----
            // Walking such a window 256 answers a step, every workgroup on its own, was 128 dependent steps of uncached
            // loads (0.16 ms a window, 128 workgroups x 512 KB of them: 4 MiB of unrelated bytes 37 ms of evaluation
            // beside 1.6 ms of searches).  The steps only hand on (M, cnt(base, M)) and cnt(base, .) -- a running maximum
            // and a running sum -- so every workgroup says what its own positions do to them (farthest match end with its
            // count, number of agreeing positions), reads the others' two words, knows the state its positions start
            // from, evaluates them exactly as a step of the loop above would, and says what it found; the first
            // workgroup that found a break or a stop point is the window's.  Same integer decisions everywhere, as before.
            {
                const int G = n_groups;
                const int64_t my0 = (int64_t)bid * kAsThreads;            // this workgroup's first position of the window
                const bool have = ln_live;
                bool stop = have && ln_l < 0;
                const int64_t l = stop ? -1 : ln_l, p = ln_p;
                const int t = (int)my0 + tid;                               // this thread's position of the window
                const unsigned long long kTagMask = (1ull << 62) - 1;
                auto put2 = [&](int64_t slot, unsigned long long a0, unsigned long long a1) {
                    __hip_atomic_store(&ans_w[2 * slot], (tag_w << 62) | a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ans_w[2 * slot + 1], (tag_w << 62) | a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                };
                // A. agreeing positions of the chunk (prefix inside it), its farthest match end with the count up to it
                const uint32_t a = (have && agree(i + t)) ? 1u : 0u;
                const uint32_t incl = wave_incl_sum(a);
                if (lane == kWave - 1) w_u32[wv] = incl;
                __syncthreads();
                uint32_t pl = incl - a, chunk_agree = 0;                    // cnt(chunk start, position); cnt over the chunk
                for (int q = 0; q < kAsWaves; ++q) { if (q < wv) pl += w_u32[q]; chunk_agree += w_u32[q]; }
                // (lane answers are capped: an end lies at most kAsCap behind its position)
                uint64_t krel = (!have || stop) ? 0ull : ((uint64_t)(uint32_t)(t + l + 1) << 32) | (uint64_t)(0xffffffffu - (pl + (uint32_t)ln_cw));
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) { const uint64_t tk = __shfl_xor(krel, o, kWave); krel = tk > krel ? tk : krel; }
                if (lane == 0) w_e[wv] = (int64_t)krel;
                __syncthreads();
                if (tid == 0) {
                    uint64_t ck = 0;
                    for (int q = 0; q < kAsWaves; ++q) ck = (uint64_t)w_e[q] > ck ? (uint64_t)w_e[q] : ck;
                    // (end + 1 in 17 bits, the count's complement in 32)
                    put2(bid, ((ck >> 32) << 32 | (ck & 0xffffffffull)) & kTagMask, (unsigned long long)chunk_agree);
                }
                lap(t_eval);
                // B. everybody's two words
                if (tid < G) {
                    unsigned long long v, v2;
                    fetch(tid, &v, &v2);
                    sm_key[tid] = v & kTagMask;
                    sm_agree[tid] = (uint32_t)(v2 & kTagMask);
                }
                __syncthreads();
                lap(t_wait);
                if (s_err) { failed = true; }
                int64_t Min = M, Cin = C, Mall = M, Call = C;              // state in front of this chunk / behind the window
                uint32_t before = 0, total = 0;
                if (!failed) {
                    for (int v = 0; v < G; ++v) {
                        if (v == bid) { Min = Mall; Cin = Call; before = total; }
                        const unsigned long long k = sm_key[v];
                        if (k) {
                            const int64_t e = i + (int64_t)(k >> 32) - 1;
                            if (e > Mall) { Mall = e; Call = S + (int64_t)total + (int64_t)(0xffffffffu - (uint32_t)k); }
                        }
                        total += sm_agree[v];
                    }
                }
                // this chunk, exactly as a step of the loop over a wave window evaluates 256 positions
                const int64_t Sj = S + (int64_t)before + (int64_t)pl;
                const int64_t e = (have && !stop) ? i + t + l : -1, ce = Sj + ln_cw;
                const uint64_t key = e < 0 ? 0ull : (((uint64_t)(uint32_t)(e - i + 1)) << 32) | (uint64_t)(0xffffffffu - (uint32_t)ce);
                uint64_t pk = key;
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const uint64_t tk = __shfl_up(pk, o, kWave);
                    if (lane >= o && tk > pk) pk = tk;
                }
                auto end_of = [&](uint64_t k) -> int64_t { return k ? (int64_t)(k >> 32) - 1 + i : (int64_t)-1; };
                auto cnt_of = [&](uint64_t k) -> int64_t { return k ? (int64_t)(0xffffffffu - (uint32_t)k) : (int64_t)0; };
                const int64_t pe = end_of(pk), pc = cnt_of(pk);
                __syncthreads();                           // (w_e was the chunk maximum's)
                if (lane == kWave - 1) { w_e[wv] = pe; w_c[wv] = pc; }
                __syncthreads();
                int64_t ce_in = Min, cc_in = Cin;          // what comes in from the left of this wave
                for (int q = 0; q < wv; ++q) if (w_e[q] > ce_in) { ce_in = w_e[q]; cc_in = w_c[q]; }
                const uint64_t xk = __shfl_up(pk, 1, kWave);
                int64_t xe = end_of(xk), xc = cnt_of(xk);
                if (lane == 0 || xe <= ce_in) { xe = ce_in; xc = cc_in; }
                int64_t Mj = xe, Cj = xc;
                if (e > Mj) { Mj = e; Cj = ce; }
                bool brk_here = false;
                int64_t carried = 0;
                if (have && !stop) {
                    carried = Cj - Sj;
                    brk_here = (l == carried && l != 0) || l > carried + 8;
                }
                const uint64_t mb = __ballot(brk_here), ms = __ballot(have && stop);
                const int32_t t0w = (int32_t)my0 + (wv << 6);
                const int32_t fb = mb ? t0w + __builtin_ctzll(mb) : 0x7fffffff, fs = ms ? t0w + __builtin_ctzll(ms) : 0x7fffffff;
                if (lane == 0) { w_brk[wv] = fb; w_stp[wv] = fs; }
                __syncthreads();
                int32_t b = 0x7fffffff, s2 = 0x7fffffff;
                for (int q = 0; q < kAsWaves; ++q) { b = w_brk[q] < b ? w_brk[q] : b; s2 = w_stp[q] < s2 ? w_stp[q] : s2; }
                const int my_brk = b < s2 ? b : -1, my_stp = (b >= s2 && s2 != 0x7fffffff) ? s2 : -1;
                const int at = my_brk >= 0 ? my_brk : my_stp;
                // the thread at the break / stop point -- or, if the chunk has none, at its last position -- says it
                const int last_t = (int)(c - 1 < my0 + kAsThreads - 1 ? c - 1 : my0 + kAsThreads - 1);
                if (!failed && (at >= 0 ? t == at : (t == last_t || (last_t < (int)my0 && tid == 0)))) {
                    const unsigned long long lf = l < 0 ? kAsStopLen : (unsigned long long)l, pf = l < 0 ? 0ull : (unsigned long long)p;
                    const unsigned long long type = my_brk >= 0 ? 1ull : my_stp >= 0 ? 2ull : 0ull;
                    put2(G + 3 * (int64_t)bid, have ? (lf << 31) | pf : 0ull,
                         (type << 56) | ((unsigned long long)(at >= 0 ? at - (int)my0 : 0) << 40) | (unsigned long long)(at >= 0 ? carried : 0));
                    put2(G + 3 * (int64_t)bid + 1, (unsigned long long)(xe - i + 1), (unsigned long long)xc);
                    put2(G + 3 * (int64_t)bid + 2, (unsigned long long)before + pl, 0ull);
                }
                lap(t_eval);
                // C. everybody's findings: the first workgroup with a break or a stop point has the window's
                if (tid < G && !failed) {
                    unsigned long long v, v2;
                    fetch(G + 3 * (int64_t)tid, &v, &v2);
                    sm_res[tid][0] = v & kTagMask; sm_res[tid][1] = v2 & kTagMask;
                    fetch(G + 3 * (int64_t)tid + 1, &v, &v2);
                    sm_res[tid][2] = v & kTagMask; sm_res[tid][3] = v2 & kTagMask;
                    fetch(G + 3 * (int64_t)tid + 2, &v, &v2);
                    sm_res[tid][4] = v & kTagMask;
                }
                __syncthreads();
                lap(t_wait);
                if (s_err) failed = true;
                if (!failed) {
                    int first = -1;
                    for (int v = 0; v < G; ++v) if ((sm_res[v][1] >> 56) != 0) { first = v; break; }
                    if (first >= 0) {
                        const unsigned long long r1 = sm_res[first][1];
                        const int where = first * kAsThreads + (int)((r1 >> 40) & 0xffffull);
                        if ((r1 >> 56) == 1ull) brk = where; else stp = where;
                        b_carried = (int64_t)(r1 & ((1ull << 40) - 1));
                        const unsigned long long r0 = sm_res[first][0];
                        b_len = (int64_t)((r0 >> 31) & 0x7fffffffull);
                        b_pos = (int64_t)(r0 & 0x7fffffffull);
                        Mrun = (int64_t)sm_res[first][2] - 1 + i;
                        Crun = (int64_t)sm_res[first][3];
                        stp_agree = (int64_t)sm_res[first][4];
                    } else {
                        Mrun = Mall; Crun = Call;
                        win_agree = (int64_t)total;
                        const unsigned long long r0 = sm_res[(c - 1) / kAsThreads][0];
                        last_len = (int64_t)((r0 >> 31) & 0x7fffffffull);
                        last_pos = (int64_t)(r0 & 0x7fffffffull);
                    }
                }
                __syncthreads();                           // (sm_*, w_* are reused)
            }
            }

            lap(t_eval);
            if (failed) break;
            if (brk >= 0) {
                const int64_t j = i + brk;
                n_search += (unsigned long long)(j - base + 1);
                cursor = j; hit_pos = b_pos; hit_len = b_len; carried_at = b_carried;
                found = true;
                window_done();
                break;
            }
            if (stp < 0) {                                 // the whole window went by (last_pos / last_len: its last answer)
                any_search = true;
                window_done();
                M = Mrun; C = Crun;
                S += win_agree;
                i += c;
                // a long differing stretch: after kAsWaveWins windows of one position per wave, one per lane (an edit of a
                // few hundred bytes is walked over by the cheaper wave windows; unrelated data by 8192 positions a step)
                if (++passed >= wave_wins) lane_mode = true;
            } else {
                // ---- 3. the stop point, on its own: searched again without the cap (by the first wave of every workgroup) ----
                window_done();                             // (nothing below reads the window's answers)
                ++n_stop;
                const int64_t j = i + stp;
                if (wv == 0) {
                    int64_t p2 = 0, l2 = 0;
                    ms_search_wave<IdxT>(old, n, sa, nw, m, j, 0, ptab, pk, &p2, &l2);
                    const int64_t cw2 = wave_cw(j, p2, l2, nullptr);       // (exactly)
                    if (lane == 0) { s_v[0] = p2; s_v[1] = l2; s_v[2] = cw2; }
                }
                __syncthreads();
                const int64_t p = s_v[0], l = s_v[1], cw = s_v[2];
                __syncthreads();
                lap(t_stop);
                const int64_t Sj = S + stp_agree;
                M = Mrun; C = Crun;
                if (j + l > M) { M = j + l; C = Sj + cw; }
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
                passed = 0;
            }
            if (M < i) {                                   // (M >= i - 1 always: the last position walked over ends at or behind itself)
                C += agree(M) ? 1 : 0;
                M = i;
            }
        }
        if (failed || left_mid) break;
        bool emit = true;
        if (!found) {
            n_search += (unsigned long long)(m - base > 0 ? m - base : 0);
            cursor = m;
            if (any_search) { hit_pos = last_pos; hit_len = last_len; }
        } else if (hit_len == carried_at) {
            emit = false;                                  // the old alignment explains it: keep scanning behind it
        }
        // (the host works on the pairs while the scan goes on: it polls the slots in order; position and match
        // travel in ONE 64-bit store, which is also the slot's "filled" mark -- no entry is ever ~0; the count of
        // Search calls beside it is read only once the launch has left the stream)
        if (bid == 0 && tid == 0) {
            __hip_atomic_store(&cum[nrec], n_search, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&rec[nrec], emit ? ((unsigned long long)cursor << 32) | (unsigned long long)(uint32_t)hit_pos
                                                : kAsSilent | ((unsigned long long)cursor << 32),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        ++nrec;
        if (cursor >= stop_at) ++past;
        if (emit) shift = hit_pos - cursor;
    }
    if (bid == 0 && tid == 0) {
        // (to the pinned result block, not to ctl: a workgroup that starts late still reads what the launch began with)
        AnchorCtl r = *ctl;
        r.nrec = (unsigned long long)nrec;
        r.cursor = cursor; r.hit_len = hit_len; r.hit_pos = hit_pos; r.shift = shift;
        r.done = (!failed && !left_mid && cursor >= m) ? 1 : 0;
        r.searches = n_search; r.windows = n_win; r.stops = n_stop;
        r.error = (failed || others_gave_up()) ? 1u : 0u;
        r.t_search = t_search; r.t_wait = t_wait; r.t_eval = t_eval; r.t_stop = t_stop;
        r.t_begin = t_begin; r.t_end = wall_clock64();
        r.mid = left_mid ? 1 : 0;
        r.base = mid_v[0]; r.i = mid_v[1]; r.M = mid_v[2]; r.C = mid_v[3]; r.S = mid_v[4];
        r.last_pos = mid_v[5]; r.last_len = mid_v[6];
        r.lane_mode = mid_f[0]; r.streak = mid_f[1]; r.passed = mid_f[2]; r.any_search = mid_f[3];
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&r);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(out);
        static_assert(sizeof(AnchorCtl) % 8 == 0, "copied as words");
        for (size_t q = 0; q < sizeof(AnchorCtl) / 8; ++q) __hip_atomic_store(&dst[q], src[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // (the launch's number last, after everything this thread has written to the host: list, counts, result)
        __hip_atomic_store(&dst[31], ln.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace dq
