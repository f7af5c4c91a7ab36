// dq_runs.h -- runs of one byte (zero padding, tables of real binaries) decided at once instead of in
// log2(run length) doubling rounds.
//
// Every suffix inside a run of byte c ties with all the others that still have h of its bytes ahead, round after
// round: the rank h bytes further on lies in the same run.  But the order among the suffixes that start with a run
// of c is known from the run alone.  Let r = the number of c's the suffix starts with and b = the byte behind them
// (or the end of the text, which sorts before every byte):
//     two such suffixes with r1 < r2 differ at offset r1, where the first has b1 and the second still has c:
//     the first is smaller iff b1 < c.
// So the suffixes whose run is followed by something SMALLER come first, by increasing r, then the others by
// decreasing r; suffixes with equal (b ≷ c, r) share exactly r bytes and are ordered by the rank of the suffix behind
// the run.  (The reference meets the same structure in its tandem-repeat shortcuts, TrSort.cs:1008-1142 tr_copy /
// tr_partialcopy.)  Here:
//
//   runlen_*_kernel     RL[i] = number of equal bytes the text has from position i on (>= 1): per 4096-byte chunk the
//                       run lengths inside the chunk (reverse scan of (length, open) pairs), one workgroup that
//                       carries runs across chunk boundaries (the same scan over chunks), and the final pass
//   run-order round     one extra round at the depth h where doubling starts: key2 = run_order_key() for the groups
//                       whose members have RL >= h (a group is uniform in that: its members share h bytes), 0 for
//                       the others (no split).  Afterwards every such group is uniform in (b ≷ c, r)
//   every later round   a member with RL[s] >= h takes the rank RL[s] bytes further on instead of h: behind its run
//
// Used for int32 indices when text_hist_kernel has seen a run of >= 64 equal bytes (DQ_RUNS=0/1 overrides).
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kRunChunk = 4096;                      // bytes per workgroup of the chunk passes
constexpr int kRunThreads = 256;
constexpr int kRunPer = kRunChunk / kRunThreads;     // 16 bytes per thread
constexpr int kRunScanThreads = 1024;

// key2 of the run-order round for a suffix s that starts with r >= h bytes c (r = RL[s]).  With a period p > 1 (the late
// run rounds, see below) the "run" is a stretch that repeats itself every p bytes, r its length, and the byte that ends
// it is compared with the byte the repetition would have put there: text[e - p] (p = 1: any byte of the run).
__device__ __forceinline__ uint32_t run_order_key(const uint8_t *__restrict__ text, int64_t n, int64_t s, uint32_t r, int period = 1)
{
    const int64_t e = s + (int64_t)r;
    const bool down = e >= n || text[e] < text[e - period];   // the end of the text sorts before every byte
    return down ? r : (0x80000000u | (0x7fffffffu - r));
}

// Reverse inclusive scan of affine maps x -> a + (open ? x : 0) over the threads of a workgroup: on return
// a[t] = f_t(f_{t+1}(... f_{T-1}(0))).  a / open: LDS arrays of T entries filled by the caller (a barrier in between).
template <int T>
__device__ __forceinline__ void affine_rscan(uint32_t *a, uint8_t *open, int t)
{
#pragma unroll
    for (int d = 1; d < T; d <<= 1) {
        uint32_t add = 0;
        uint8_t o = 0;
        const bool take = open[t] && t + d < T;
        if (take) { add = a[t + d]; o = open[t + d]; }
        __syncthreads();
        if (take) { a[t] += add; open[t] = o; }
        __syncthreads();
    }
}

// pass 1 (kFinal = false): per chunk, lead[c] = run length at the chunk's first byte counted inside the chunk,
//         link[c] = the run reaches the chunk's end AND the text goes on with the same byte
// pass 3 (kFinal = true):  RL[i] for every position, with carry[c + 1] = true run length at the first byte of the next chunk
// period p (1 for runs of one byte): position x is LINKED to x + 1 iff text[x] == text[x + p] (and x + p < n); a run is a
// maximal chain of links, its length in positions r, and what is stored is how far the text goes on repeating itself
// from there: RL = min(r - 1 + p, n - position).  For p = 1 that is the number of equal bytes, as before.
template <bool kFinal>
__global__ __launch_bounds__(kRunThreads) void runlen_chunk_kernel(const uint8_t *__restrict__ text, int64_t n,
                                                                   uint32_t *__restrict__ lead, uint8_t *__restrict__ link,
                                                                   const uint32_t *__restrict__ carry, uint32_t *__restrict__ RL,
                                                                   int period = 1)
{
    __shared__ uint32_t s_a[kRunThreads];
    __shared__ uint8_t s_open[kRunThreads];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * kRunChunk;
    const int64_t end = c0 + kRunChunk < n ? c0 + kRunChunk : n;           // chunk = [c0, end)
    const int64_t p0 = c0 + (int64_t)t * kRunPer;
    auto linked = [&](int64_t x) -> bool { return x + period < n && text[x] == text[x + period]; };
    bool e[kRunPer];                                                      // e[i]: position p0 + i is linked to the next one
#pragma unroll
    for (int i = 0; i < kRunPer; ++i) e[i] = linked(p0 + i);
    const bool end_linked = end < n && linked(end - 1);                    // the chunk's last position, to the next chunk
    // run lengths inside my segment, from the right; a position beyond `end` counts as a break
    uint32_t in[kRunPer];
    uint32_t run = 0;
#pragma unroll
    for (int i = kRunPer - 1; i >= 0; --i) {
        const bool valid = p0 + i < end;
        const bool cont = valid && i + 1 < kRunPer && p0 + i + 1 < end && e[i];
        run = valid ? (cont ? run + 1 : 1) : 0;
        in[i] = run;
    }
    // my segment's map: its leading run, open iff that run covers the whole segment and is linked to the next segment
    // (inside the chunk)
    const int64_t seg_end = p0 + kRunPer;
    const bool full = p0 < end && in[0] == (uint32_t)kRunPer && seg_end < end && e[kRunPer - 1];
    s_a[t] = p0 < end ? in[0] : 0;
    s_open[t] = full ? 1 : 0;
    __syncthreads();
    affine_rscan<kRunThreads>(s_a, s_open, t);
    // s_a[t] = run length at my first byte, counted inside the chunk
    if (!kFinal) {
        if (t == 0) {
            lead[blockIdx.x] = s_a[0];
            link[blockIdx.x] = (c0 + (int64_t)s_a[0] == end && end_linked) ? 1 : 0;
        }
        return;
    }
    const uint32_t next_seg = t + 1 < kRunThreads ? s_a[t + 1] : 0;          // run length at the next segment's first byte
    const uint32_t next_chunk = end < n ? carry[blockIdx.x + 1] : 0;         // ... at the next chunk's first byte (true length)
#pragma unroll
    for (int i = 0; i < kRunPer; ++i) {
        const int64_t p = p0 + i;
        if (p >= end) break;
        uint32_t r = in[i];
        // the run reaches my segment's end and goes on in the next segment (same chunk)?
        if (i + (int)r == kRunPer && seg_end < end && e[kRunPer - 1]) r += next_seg;
        // ... reaches the chunk's end and goes on in the next chunk?
        if (p + (int64_t)r == end && end_linked) r += next_chunk;
        const int64_t far = (int64_t)r - 1 + period, left = n - p;
        RL[p] = (uint32_t)(far < left ? far : left);
    }
}

// pass 2: carry[c] = true run length at the first byte of chunk c = lead[c] + (link[c] ? carry[c + 1] : 0); one workgroup
static __global__ __launch_bounds__(kRunScanThreads) void runlen_carry_kernel(const uint32_t *__restrict__ lead,
                                                                       const uint8_t *__restrict__ link, int64_t nchunks,
                                                                       uint32_t *__restrict__ carry /*[nchunks + 1]*/)
{
    __shared__ uint32_t s_a[kRunScanThreads];
    __shared__ uint8_t s_open[kRunScanThreads];
    const int t = threadIdx.x;
    const int64_t per = (nchunks + kRunScanThreads - 1) / kRunScanThreads;
    const int64_t lo = (int64_t)t * per, hi = lo + per < nchunks ? lo + per : nchunks;
    // my range as one map, composed from the right
    uint32_t a = 0;
    uint8_t open = 1;
    for (int64_t c = hi - 1; c >= lo; --c) {
        a = lead[c] + (link[c] ? a : 0);
        open = link[c] ? open : 0;
    }
    if (lo >= hi) { a = 0; open = 1; }                       // an empty range passes its input through
    s_a[t] = a;
    s_open[t] = open;
    __syncthreads();
    affine_rscan<kRunScanThreads>(s_a, s_open, t);
    // the value entering my range from the right = the scanned value of the next thread
    uint32_t x = t + 1 < kRunScanThreads ? s_a[t + 1] : 0;
    for (int64_t c = hi - 1; c >= lo; --c) {
        x = lead[c] + (link[c] ? x : 0);
        carry[c] = x;
    }
    if (t == 0) carry[nchunks] = 0;
}

}  // namespace dq
