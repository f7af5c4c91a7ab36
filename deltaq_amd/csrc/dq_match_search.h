// dq_match_search.h -- Diff.Create's match search on the device-resident suffix array.
//
// Reference: src/DeltaQ.BsDiff/Diff.cs:267-298  Search(I, oldData, newData[scan..], 0, oldData.Length, out pos)
//            (CompareBytes :244-246 = SequenceCompareTo, MatchLength :248-265), called from the scan loop at :106.
//
// What Search returns is fixed by the text alone, not by its probe sequence: SequenceCompareTo is a total
// order, so "suffix I[mid] < query" is monotone over the suffix array and the loop ends with
//     start = max(g - 1, 0),  end = start + 1,     g = number of suffixes of old smaller than the query,
// (I[n] is the zeroed sentinel slot of Diff.cs:78, i.e. suffix 0), and the answer is the longer of the two
// match lengths, ties to `end`.  Any way of finding g is bit-exact; this kernel uses a lower-bound search
// that skips the prefix both interval ends are known to share with the query (min(llcp, rlcp)), so a query
// costs O(match length + log n) byte comparisons instead of O(match length * log n).
//
//   match_search_kernel   one query per lane.  Comparisons of up to kMsLaneBytes bytes beyond the known
//                         common prefix are done by the lane itself, 8 bytes a step (in the regions where old
//                         and new differ -- where the scan loop calls Search at every byte -- nearly all are a
//                         few bytes long); longer ones are handed to the whole wave, one at a time: 64 lanes
//                         x 8 bytes per step, first mismatch by ballot.  Random access to the SA (w B per probe) and to
//                         old (one 64-B sector per probe): latency-bound, ~log2(n) dependent probes.
//   cap > 0               a query whose comparison runs more than `cap` bytes past the known prefix is given
//                         up (len = -1): speculative batches inside a long match must not cost O(match
//                         length) each; the caller repeats that one position with cap = 0.
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kMsThreads = 256;
constexpr int kMsLaneBytes = 256;        // a lane compares this far on its own (8 bytes a step) before the wave takes over
constexpr int kMsWaveLaneBytes = 64;     // ... in the one-query-per-wave search, whose caller waits for the slowest wave

__device__ __forceinline__ uint64_t ms_readlane64(uint64_t v, int lane)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
}

// 8 bytes at p (any alignment) from three aligned dwords; every dword touched holds at least one of the 12
// bytes p[0..11], which the caller guarantees to be inside the buffer
__device__ __forceinline__ uint64_t ms_load8(const uint8_t *p)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    const uint32_t sh = (uint32_t)(a & 3);
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
    return ((uint64_t)hi << 32) | lo;
}

// Length of the common prefix of a[0..la) and b[0..lb), given that the first k bytes are equal.  Called by
// the WHOLE wave with wave-uniform arguments.
__device__ __forceinline__ int64_t ms_wave_lcp(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb, int64_t k)
{
    const int lane = lane_id();
    const int64_t lim = la < lb ? la : lb;
    // 4 KiB per step (eight independent 512-byte slices in flight: a long match is a chain of dependent steps, and the
    // scan loop waits for it), then 2 KiB, while 12 more bytes exist beyond every lane's 8.  (Sixteen slices were
    // measured: the wave search spills, every window twice as slow.)
    while (k + 8 * 64 * 8 + 4 <= lim) {
        uint64_t x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t j = k + u * 512 + 8 * lane;
            x[u] = ms_load8(a + j) ^ ms_load8(b + j);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t bad = __ballot(x[u] != 0);
            if (bad) {
                const int f = __builtin_ctzll(bad);
                const uint64_t xf = ms_readlane64(x[u], f);
                return k + u * 512 + 8 * f + (__builtin_ctzll(xf) >> 3);
            }
        }
        k += 8 * 64 * 8;
    }
    while (k + 4 * 64 * 8 + 4 <= lim) {
        uint64_t x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = k + u * 512 + 8 * lane;
            x[u] = ms_load8(a + j) ^ ms_load8(b + j);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t bad = __ballot(x[u] != 0);
            if (bad) {
                const int f = __builtin_ctzll(bad);
                const uint64_t xf = ms_readlane64(x[u], f);
                return k + u * 512 + 8 * f + (__builtin_ctzll(xf) >> 3);
            }
        }
        k += 4 * 64 * 8;
    }
    // 512 bytes per step
    while (k + 64 * 8 + 4 <= lim) {
        const int64_t j = k + 8 * lane;
        const uint64_t x = ms_load8(a + j) ^ ms_load8(b + j);
        const uint64_t bad = __ballot(x != 0);
        if (bad) {
            const int f = __builtin_ctzll(bad);
            const uint64_t xf = ms_readlane64(x, f);
            return k + 8 * f + (__builtin_ctzll(xf) >> 3);
        }
        k += 64 * 8;
    }
    // the rest byte by byte, 64 at a time
    while (k < lim) {
        const int64_t j = k + lane;
        const bool diff = j < lim && a[j] != b[j];
        const uint64_t bad = __ballot(diff);
        if (bad) return k + __builtin_ctzll(bad);
        k += 64;
    }
    return lim;
}

// [ptab[v], ptab[v + 1]) holds the suffixes s with pattern(v) <= s < pattern(v + 1): the ones that start with the pk
// bytes of v -- and, at its upper end, up to pk - 1 suffixes SHORTER than pk bytes that sort between the two patterns
// without sharing them (old ending in the byte c puts the one-byte suffix "c" into the range of v = (c - 1, 255):
// "c-1 255 ..." < "c" < "c 0").  The search below starts its comparisons behind the pk shared bytes, so those are cut
// off first; they are larger than every suffix that does start with v, hence larger than the query.
template <typename IdxT>
__device__ __forceinline__ void ms_trim_short_suffixes(const IdxT *__restrict__ sa, int64_t n, int pk, int64_t L, int64_t *R)
{
    for (int t = 1; t < pk && *R > L; ++t) {
        if (n - (int64_t)sa[*R - 1] < pk) --*R; else break;
    }
}

// prefix_bounds_kernel: ptab[v] = number of suffixes of old below the pk-byte string with big-endian value v, for
// v = 0 .. 256^pk (ptab[256^pk] = n); one thread per v, the same lower bound as below with a pk-byte pattern.
// A query of >= pk bytes with prefix value v then has its lower bound inside [ptab[v], ptab[v + 1]], and every
// suffix inside that range -- but for the short ones ms_trim_short_suffixes() removes -- shares those pk bytes with
// it: the search starts ~8 * pk probes further down.
// Built once per old file by the scan-loop driver (dq_diff.hip::SearchWindows), where a Search is one
// dependent round trip to the device and its ~log2(n) probes of ~1 us each are what the round trip costs.
// coarse: the table of pk - 1 bytes, if it has been built: the bound of v lies inside [coarse[v >> 8], coarse[(v >> 8) + 1]]
// (the suffixes below the shorter pattern are below every pattern it begins, those below v are below the next shorter
// pattern) -- 8 probes instead of log2(n): the 3-byte table of a 16 MiB file 0.84 -> 0.3 ms, a twelfth of a Diff.Create.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void prefix_bounds_kernel(const uint8_t *__restrict__ old, int64_t n,
                                                               const IdxT *__restrict__ sa, int pk, IdxT *__restrict__ ptab,
                                                               const IdxT *__restrict__ coarse = nullptr)
{
    const int64_t total = 1ll << (8 * pk);
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v > total) return;
    if (v == total) { ptab[v] = (IdxT)n; return; }
    int64_t L = 0, R = n;
    if (coarse) { L = (int64_t)coarse[v >> 8]; R = (int64_t)coarse[(v >> 8) + 1]; }
    while (L < R) {
        const int64_t mid = L + ((R - L) >> 1);
        const int64_t p = (int64_t)sa[mid], la = n - p;
        bool less = la < pk;                               // all compared bytes equal: the shorter (a proper prefix) first
        for (int j = 0; j < pk && j < la; ++j) {
            const int a = old[p + j], b = (int)((v >> (8 * (pk - 1 - j))) & 0xff);
            if (a != b) { less = a < b; break; }
        }
        if (less) L = mid + 1; else R = mid;
    }
    ptab[v] = (IdxT)L;
}

// One query per lane: Search(I, old, nw[scan..], 0, n).  Called by whole waves (the wave-wide comparison needs
// every lane); `live` = this lane has a query.  *len_res = -1 when the cap was hit.
template <typename IdxT>
__device__ __forceinline__ void ms_search_one(const uint8_t *__restrict__ old, int64_t n, const IdxT *__restrict__ sa,
                                              const uint8_t *__restrict__ nw, int64_t m, int64_t scan, bool live, int64_t cap,
                                              const IdxT *__restrict__ ptab, int pk, int64_t *pos_res, int64_t *len_res)
{
    const int lane = lane_id();
    const uint8_t *q = nw + scan;
    const int64_t lq = live ? m - scan : 0;

    // lcp(query, suffix at old position p), given k equal leading bytes: per lane up to kMsLaneBytes, then
    // the wave takes over.  Every lane of the wave calls this the same number of times (the search loops
    // below are padded with idle rounds), so the hand-over loop always has all 64 lanes.
    bool gave_up = false;
    auto lcp_with = [&](int64_t p, int64_t k, bool want) -> int64_t {
        const uint8_t *a = old + p;
        const int64_t la = n - p;
        const int64_t lim = la < lq ? la : lq;
        int64_t j = k;
        bool more = false;
        if (want) {
            const int64_t stop = j + kMsLaneBytes < lim ? j + kMsLaneBytes : lim;
            // 8 bytes a step while 12 bytes exist in both buffers (ms_load8 touches whole dwords), then byte by byte
            bool diff = false;
            while (j + 8 <= stop && j + 12 <= la && j + 12 <= lq) {
                const uint64_t x = ms_load8(a + j) ^ ms_load8(q + j);
                if (x) { j += __builtin_ctzll(x) >> 3; diff = true; break; }
                j += 8;
            }
            if (!diff) while (j < stop && a[j] == q[j]) ++j;
            more = j == stop && stop < lim;
            if (more && cap > 0 && j - k >= cap) { gave_up = true; more = false; }
        }
        uint64_t need = __ballot(more);
        while (need) {
            const int f = __builtin_ctzll(need);
            need &= need - 1;
            const uint8_t *fa = reinterpret_cast<const uint8_t *>(ms_readlane64(reinterpret_cast<uint64_t>(a), f));
            const uint8_t *fq = reinterpret_cast<const uint8_t *>(ms_readlane64(reinterpret_cast<uint64_t>(q), f));
            const int64_t fla = (int64_t)ms_readlane64((uint64_t)la, f), flq = (int64_t)ms_readlane64((uint64_t)lq, f);
            const int64_t fj = (int64_t)ms_readlane64((uint64_t)j, f);
            int64_t lim2 = fla < flq ? fla : flq;
            const int64_t fk = (int64_t)ms_readlane64((uint64_t)k, f);
            bool capped = false;
            if (cap > 0 && lim2 > fk + cap) { lim2 = fk + cap; capped = true; }       // (wave-uniform: cap is)
            const int64_t r = ms_wave_lcp(fa, lim2, fq, lim2, fj);
            if (lane == f) {
                j = r;
                if (capped && r == lim2) gave_up = true;
            }
        }
        return j;
    };
    // suffix at p < query, from their lcp
    auto less_than_query = [&](int64_t p, int64_t l) -> bool {
        const int64_t la = n - p;
        if (l == la || l == lq) return la < lq;                  // one is a prefix of the other: the shorter first
        return old[p + l] < q[l];
    };

    // ---- g = number of suffixes smaller than the query: lower bound over [L, R) ----
    int64_t L = 0, R = live ? n : 0;
    // lcp with SA[L-1] / SA[R] once they have been probed (l_known / r_known); until then a lower bound of the lcp
    // with everything inside [L, R)
    int64_t llcp = 0, rlcp = 0;
    bool l_known = false, r_known = false;
    if (ptab && live && lq >= pk) {
        int64_t v = 0;
        for (int j = 0; j < pk; ++j) v = (v << 8) | q[j];
        L = (int64_t)ptab[v];
        R = (int64_t)ptab[v + 1];
        ms_trim_short_suffixes(sa, n, pk, L, &R);
        llcp = rlcp = pk;
    }
    for (;;) {
        const bool active = L < R && !gave_up;
        if (!__any(active)) break;
        int64_t mid = 0, p = 0;
        if (active) {
            mid = L + ((R - L) >> 1);
            p = (int64_t)sa[mid];
        }
        const int64_t l = lcp_with(p, llcp < rlcp ? llcp : rlcp, active);
        if (active && !gave_up) {
            if (less_than_query(p, l)) { L = mid + 1; llcp = l; l_known = true; }
            else { R = mid; rlcp = l; r_known = true; }
        }
    }
    const int64_t g = L;

    // ---- the two candidates: I[start], I[end] with start = max(g - 1, 0), end = start + 1, I[n] = 0 ----
    const int64_t start = g > 0 ? g - 1 : 0, end = start + 1;
    const bool usable = live && !gave_up && n > 0;
    const int64_t ps = usable ? (int64_t)sa[start] : 0;
    const int64_t pe = usable && end < n ? (int64_t)sa[end] : 0;
    // x: lcp with I[start].  g >= 1: SA[g-1], the last probe that moved L (llcp); g == 0: SA[0] = SA[R] (rlcp) --
    // if that end of the interval was probed at all (with a prefix table the search may start right on it)
    const bool x_known = g > 0 ? l_known : r_known;
    const int64_t x_meas = lcp_with(ps, 0, usable && !x_known);
    const int64_t x = x_known ? (g > 0 ? llcp : rlcp) : x_meas;
    // y: lcp with I[end].  g >= 1 and g < n: SA[g] = SA[R] if it was probed (rlcp); otherwise it has to be measured
    const bool y_known = g > 0 && g < n && r_known;
    const int64_t y_meas = lcp_with(pe, 0, usable && !y_known);
    const int64_t y = y_known ? rlcp : y_meas;
    if (gave_up) { *pos_res = 0; *len_res = -1; }
    else if (n == 0) { *pos_res = 0; *len_res = 0; }                              // I = { 0 }: both candidates are I[0]
    else if (x > y) { *pos_res = ps; *len_res = x; }
    else { *pos_res = pe; *len_res = y; }
}

// One query per WAVE (the scan-loop driver's windows: a few hundred queries, and the launch is one dependent
// round trip of the host loop, so what counts is the depth of the probe chain, not the work).  A 65-ary lower
// bound: the 64 lanes probe 64 split points of [L, R) at once (every entry once the interval has <= 64), each
// lane comparing its suffix with the query from the prefix the interval is known to share; "suffix < query" is
// monotone over the lanes, so the number of lanes that say yes is the new interval.  ~log65(R - L) levels instead of
// log2: 3 for a range of 1e5 suffixes.  cap as in ms_search_one; the answers are the same by construction (g is g).
template <typename IdxT>
__device__ __forceinline__ void ms_search_wave(const uint8_t *__restrict__ old, int64_t n, const IdxT *__restrict__ sa,
                                               const uint8_t *__restrict__ nw, int64_t m, int64_t scan, int64_t cap,
                                               const IdxT *__restrict__ ptab, int pk, int64_t *pos_res, int64_t *len_res,
                                               int64_t *over_at = nullptr, bool resume_first = false, bool *went_exact = nullptr)
{
    // resume_first: a comparison that runs over the cap normally ends the search (len = -1: "inside a long match").
    // If the match it ran into does NOT cover the byte before the query (old[p - 1] != new[scan - 1]), this position is
    // probably the FIRST of that match -- the one the scan loop will want exactly -- and the search goes on from where
    // it stands with the cap lifted, instead of being repeated from the top by the caller.
    const int lane = lane_id();
    const uint8_t *q = nw + scan;
    const int64_t lq = m - scan;
    bool gave_up = false;                                       // wave-uniform
    int64_t gave_up_at = -1;                                    // old position whose comparison went over the cap
    int handed = 0;                                             // lanes of the last level whose comparison the wave had to finish
    bool narrow = false;                                        // repetitive text: one probe per level from here on
    // lcp(query, suffix at p) from k equal bytes, every lane its own p: up to kMsLaneBytes alone, then one lane at a
    // time with the whole wave; *over = the cap was exceeded
    auto lcp_lanes = [&](int64_t p, int64_t k, bool want, bool *over) -> int64_t {
        const uint8_t *a = old + p;
        const int64_t la = n - p;
        const int64_t lim = la < lq ? la : lq;
        int64_t j = k;
        bool more = false;
        *over = false;
        if (want) {
            // A lane compares kMsWaveLaneBytes on its own -- one step of 32 bytes in flight together, then 8 at a time --
            // before the wave takes the comparison over (512 bytes a step and more): the window the scan loop waits for
            // is as slow as its slowest search, and that is the one position whose match goes on for kilobytes; a lane
            // walking 256 bytes of it in 32 dependent steps cost more than the rest of the search.  With a cap, never
            // beyond the cap.
            const int64_t budget = (cap > 0 && cap < kMsWaveLaneBytes) ? cap : (int64_t)kMsWaveLaneBytes;
            const int64_t stop = j + budget < lim ? j + budget : lim;
            bool diff = false;
            if (j + 32 <= stop && j + 36 <= la && j + 36 <= lq) {
                uint64_t x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = ms_load8(a + j + 8 * u) ^ ms_load8(q + j + 8 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!diff) {
                        if (x[u]) { j += __builtin_ctzll(x[u]) >> 3; diff = true; }
                        else j += 8;
                    }
                }
            }
            while (!diff && j + 8 <= stop && j + 12 <= la && j + 12 <= lq) {
                const uint64_t x = ms_load8(a + j) ^ ms_load8(q + j);
                if (x) { j += __builtin_ctzll(x) >> 3; diff = true; break; }
                j += 8;
            }
            if (!diff) while (j < stop && a[j] == q[j]) ++j;
            more = j == stop && stop < lim;
            if (more && cap > 0 && j - k >= cap) { *over = true; more = false; }
        }
        uint64_t need = __ballot(more);
        handed = __popcll(need);
        while (need) {
            const int f = __builtin_ctzll(need);
            need &= need - 1;
            const uint8_t *fa = reinterpret_cast<const uint8_t *>(ms_readlane64(reinterpret_cast<uint64_t>(a), f));
            const int64_t fla = (int64_t)ms_readlane64((uint64_t)la, f);
            const int64_t fj = (int64_t)ms_readlane64((uint64_t)j, f);
            int64_t lim2 = fla < lq ? fla : lq;
            bool capped = false;
            if (cap > 0 && lim2 > k + cap) { lim2 = k + cap; capped = true; }
            const int64_t r = ms_wave_lcp(fa, lim2, q, lim2, fj);
            if (lane == f) {
                j = r;
                if (capped && r == lim2) *over = true;
            }
        }
        return j;
    };
    auto less_than_query = [&](int64_t p, int64_t l) -> bool {
        const int64_t la = n - p;
        if (l == la || l == lq) return la < lq;
        return old[p + l] < q[l];
    };

    int64_t L = 0, R = n, llcp = 0, rlcp = 0;
    bool l_known = false, r_known = false;
    if (ptab && lq >= pk) {
        int64_t v = 0;
        for (int j = 0; j < pk; ++j) v = (v << 8) | q[j];
        L = (int64_t)ptab[v];
        R = (int64_t)ptab[v + 1];
        ms_trim_short_suffixes(sa, n, pk, L, &R);
        llcp = rlcp = pk;
    }
    while (L < R && !gave_up) {
        const int64_t span = R - L;
        // Many long comparisons in one level (every suffix around the query shares kilobytes with it: runs, periodic
        // data) would each be finished by the whole wave, one after the other: from then on a plain binary search,
        // whose single probe per level starts from the prefix both ends already share.
        const int cnt = narrow ? 1 : (span <= kWave ? (int)span : kWave);
        // split points: every entry, or 64 distinct interior points (span >= 65: consecutive ones differ by >= 1)
        const int64_t mid = narrow ? L + (span >> 1)
                                   : (span <= kWave ? L + lane : L + (int64_t)(((unsigned __int128)(uint64_t)span * (uint64_t)(lane + 1)) / 65u));
        const bool act = lane < cnt;
        const int64_t p = act ? (int64_t)sa[mid] : 0;
        bool over;
        const int64_t l = lcp_lanes(p, llcp < rlcp ? llcp : rlcp, act, &over);
        if (handed > 4) narrow = true;
        const uint64_t ov = __ballot(over);
        if (ov) {
            gave_up_at = (int64_t)ms_readlane64((uint64_t)p, __builtin_ctzll(ov));
            if (resume_first && cap > 0 && !(gave_up_at > 0 && scan > 0 && old[gave_up_at - 1] == nw[scan - 1])) {
                cap = 0;                                        // this level again, exactly; everything after it too
                if (went_exact) *went_exact = true;
                continue;
            }
            gave_up = true;
            break;
        }
        const uint64_t yes = __ballot(act && less_than_query(p, l));      // a prefix of the active lanes
        const int f = __popcll(yes);
        if (f > 0) {
            L = (int64_t)ms_readlane64((uint64_t)mid, f - 1) + 1;
            llcp = (int64_t)ms_readlane64((uint64_t)l, f - 1);
            l_known = true;
        }
        if (f < cnt) {
            R = (int64_t)ms_readlane64((uint64_t)mid, f);
            rlcp = (int64_t)ms_readlane64((uint64_t)l, f);
            r_known = true;
        }
    }
    const int64_t g = L;
    const int64_t start = g > 0 ? g - 1 : 0, end = start + 1;
    const bool usable = !gave_up && n > 0;
    const int64_t ps = usable ? (int64_t)sa[start] : 0;
    const int64_t pe = usable && end < n ? (int64_t)sa[end] : 0;
    // the two candidates' match lengths: known from the search where that end of the interval was probed, else measured
    auto measure = [&](int64_t p) -> int64_t {
        int64_t lim = (n - p) < lq ? (n - p) : lq;
        bool capped = false;
        if (cap > 0 && lim > cap) { lim = cap; capped = true; }
        int64_t r = ms_wave_lcp(old + p, lim, q, lim, 0);
        if (capped && r == lim) {
            if (resume_first && !(p > 0 && scan > 0 && old[p - 1] == nw[scan - 1])) {
                cap = 0;
                if (went_exact) *went_exact = true;
                const int64_t full = (n - p) < lq ? (n - p) : lq;
                r = ms_wave_lcp(old + p, full, q, full, r);
            } else {
                gave_up = true;
                gave_up_at = p;
            }
        }
        return r;
    };
    const bool x_known = g > 0 ? l_known : r_known;
    int64_t x = x_known ? (g > 0 ? llcp : rlcp) : 0;
    if (usable && !x_known) x = measure(ps);
    const bool y_known = g > 0 && g < n && r_known;
    int64_t y = y_known ? rlcp : 0;
    if (usable && !y_known && !gave_up) y = measure(pe);
    if (over_at) *over_at = gave_up_at;
    if (gave_up) { *pos_res = 0; *len_res = -1; }
    else if (n == 0) { *pos_res = 0; *len_res = 0; }
    else if (x > y) { *pos_res = ps; *len_res = x; }
    else { *pos_res = pe; *len_res = y; }
}

// Window kernel of the scan-loop driver: one wave per position scan0 + i; position 0 -- the one the loop is standing
// on -- exactly, the speculative ones behind it with the cap.  A capped position is where a long match lies; the
// loop will ask for the exact answer at the FIRST such position of the window (there it jumps).  A wave cannot see
// the other positions' results, but it can tell whether it is probably that first one: if the long match it ran into
// also covers the byte before (old[p-1] == new[scan-1]), the position before is inside the same match and capped
// too.  If not, the wave searches again without the cap.  (A wrong guess costs time only: a position still capped
// when the loop reaches it opens a new window there.)
// SECOND STAGE (count2 > 0, polled windows only): where the loop breaks it jumps by the length of the match it
// broke on, and its next window starts there -- one more dependent round trip.  The waves that answered exactly
// (position 0, or a re-searched first position of a long match) publish (position, length) in a device mailbox, the
// smallest position winning; count2 more waves wait for all count (<= 4096) first-stage waves, take the winner (t, len) and answer
// the window at scan0 + t + len in the same launch.  The host uses those answers only if its loop really jumps
// there (they are tagged with their start), so a wrong guess costs time, never correctness; the wait is bounded.
constexpr uint64_t kMsSkipped = 0x8000000080000001ull;           // second-stage slot: not computed
constexpr int kMsLongMatch = 32;                                 // a match the loop will break on (len > oldscore + 8)

template <typename IdxT>
__global__ __launch_bounds__(kMsThreads) void match_search_wave_kernel(
    const uint8_t *__restrict__ old, int64_t n, const IdxT *__restrict__ sa, const uint8_t *__restrict__ nw, int64_t m,
    int64_t scan0, int64_t count, int64_t cap, IdxT *__restrict__ pos_out, IdxT *__restrict__ len_out,
    const IdxT *__restrict__ ptab, int pk, uint64_t *__restrict__ packed_out /* (len << 32 | pos) per position, or null */,
    int64_t count2 = 0, uint64_t *__restrict__ packed2 = nullptr /* second stage: [0] its start, [1 + i] its answers */,
    unsigned long long *__restrict__ mail = nullptr /* [0] winner, [1] finished first-stage waves (cumulative) */,
    unsigned long long ticket = 0 /* of this launch, < 2^20, growing */, unsigned long long done_target = 0 /* mail[1] when stage 1 is through */,
    int walk_on = 0 /* second stage without a winner: answer the positions behind the window */,
    int no_resume = 0 /* 1: a capped first position is searched again from the top (the older form) */)
{
    const int64_t qi = (int64_t)blockIdx.x * (kMsThreads / kWave) + (threadIdx.x >> 6);
    if (qi >= count + count2) return;                            // (whole waves)
    int64_t base = scan0, idx = qi;                               // this wave answers position base + idx
    if (qi >= count) {
        // ---- second stage: wait for the first one, then take the winner ----
        idx = qi - count;
        unsigned long long v = 0;
        bool ready = false;
        for (int spins = 0; spins < (1 << 16); ++spins) {
            if (__hip_atomic_load(&mail[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= done_target) { ready = true; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (ready) v = __hip_atomic_load(&mail[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // winner word: ticket << 44 | (4095 - position) << 32 | length, kept by atomic max: a newer launch beats an older
        // one, inside a launch the smallest position wins (no reset between launches)
        // No winner: no position of the window met a long match -- the loop is walking through a stretch that differs
        // (an inserted / replaced run of bytes longer than the window) and will ask for the positions right behind it.
        const bool won = ready && (v >> 44) == ticket;
        if (won) base = scan0 + (int64_t)(4095 - ((v >> 32) & 0xfff)) + (int64_t)(uint32_t)v;
        else base = scan0 + count;
        const bool go = ready && (won || walk_on);
        if (!go || base + idx >= m) {
            if (lane_id() == 0) {
                if (idx == 0) __hip_atomic_store(&packed2[0], kMsSkipped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&packed2[1 + idx], kMsSkipped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        if (idx == 0 && lane_id() == 0)
            __hip_atomic_store(&packed2[0], (uint64_t)base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    int64_t pos = 0, len = 0, at = -1;
    const int64_t scan = base + idx;
    bool exact = idx == 0;
    bool went_exact = false;
    ms_search_wave<IdxT>(old, n, sa, nw, m, scan, exact ? 0 : cap, ptab, pk, &pos, &len, &at, /*resume_first=*/!no_resume, &went_exact);
    exact = exact || went_exact;
    if (no_resume && len < 0 && !(at > 0 && scan > 0 && old[at - 1] == nw[scan - 1])) {
        ms_search_wave<IdxT>(old, n, sa, nw, m, scan, 0, ptab, pk, &pos, &len);
        exact = true;
    }
    if (lane_id() == 0) {
        if (packed_out) {
            // polled by the host loop in pinned memory (a round trip of the scan loop is worth the ~10 us of
            // completion-signal latency): position and length travel in ONE 64-bit store, so that no ordering between
            // two stores is needed (two 32-bit stores with a system fence in between were seen out of order by the
            // host about once in 50 000 windows)
            __hip_atomic_store(qi < count ? &packed_out[qi] : &packed2[1 + idx], ((uint64_t)(uint32_t)(int32_t)len << 32) | (uint32_t)(int32_t)pos,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            pos_out[qi] = (IdxT)pos;
            len_out[qi] = (IdxT)len;
        }
        if (mail && qi < count) {
            if (exact && len >= kMsLongMatch)
                __hip_atomic_fetch_max(&mail[0], (ticket << 44) | ((unsigned long long)(4095 - qi) << 32) | (unsigned long long)(uint32_t)len,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&mail[1], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// exact_first (with cap > 0; the scan-loop driver's windows): the FIRST position of the window that hit the cap is
// searched again without a cap inside the same launch -- it is where the scan loop will meet the next long match
// and ask for the exact answer, and a second launch for one query costs a whole round trip to the device.
// (Workgroup 0 only: later ones would pay for long comparisons the loop never looks at.)
template <typename IdxT>
__global__ __launch_bounds__(kMsThreads) void match_search_kernel(
    const uint8_t *__restrict__ old, int64_t n, const IdxT *__restrict__ sa, const uint8_t *__restrict__ nw, int64_t m,
    const int64_t *__restrict__ scans, int64_t scan0, int64_t count, int64_t cap, IdxT *__restrict__ pos_out,
    IdxT *__restrict__ len_out, const IdxT *__restrict__ ptab = nullptr, int pk = 0, int exact_first = 0)
{
    __shared__ int s_first;
    const int64_t qi = (int64_t)blockIdx.x * kMsThreads + threadIdx.x;
    const bool live = qi < count;
    const int64_t scan = live ? (scans ? scans[qi] : scan0 + qi) : 0;
    int64_t pos = 0, len = 0;
    ms_search_one<IdxT>(old, n, sa, nw, m, scan, live, cap, ptab, pk, &pos, &len);
    if (exact_first && cap > 0 && blockIdx.x == 0) {                                  // (uniform over the workgroup)
        if (threadIdx.x == 0) s_first = kMsThreads;
        __syncthreads();
        if (live && len < 0) atomicMin(&s_first, (int)threadIdx.x);
        __syncthreads();
        const int first = s_first;
        if (first < kMsThreads) {
            const bool mine = (int)threadIdx.x == first;
            int64_t p2 = 0, l2 = 0;
            ms_search_one<IdxT>(old, n, sa, nw, m, scan, live && mine, 0, ptab, pk, &p2, &l2);
            if (mine) { pos = p2; len = l2; }
        }
    }
    if (live) { pos_out[qi] = (IdxT)pos; len_out[qi] = (IdxT)len; }
}


}  // namespace dq
