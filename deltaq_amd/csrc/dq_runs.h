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

// Reverse inclusive scan of the same maps over the kRunThreads threads of a workgroup, in registers: inside a wave by
// shuffles (no barrier), the waves' own maps through LDS.  On return a = f_t(f_{t+1}(... f_{T-1}(0))).
__device__ __forceinline__ uint32_t affine_rscan_regs(uint32_t a, bool open, uint32_t *w_a /*[waves]*/, uint8_t *w_open, int t)
{
    const int lane = t & (kWave - 1), wv = t >> 6;
    constexpr int kWaves = kRunThreads / kWave;
    uint32_t o = open ? 1u : 0u;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t add = __shfl_down(a, d, kWave);
        const uint32_t oo = __shfl_down(o, d, kWave);
        if (o && lane + d < kWave) { a += add; o = oo; }
    }
    if (lane == 0) { w_a[wv] = a; w_open[wv] = (uint8_t)o; }       // the whole wave as one map
    __syncthreads();
    uint32_t x = 0;                                                // what enters my wave from the right
    for (int v = kWaves - 1; v > wv; --v) x = w_a[v] + (w_open[v] ? x : 0u);
    return a + (o ? x : 0u);
}

// pass 1 (kFinal = false): per chunk, lead[c] = run length at the chunk's first byte counted inside the chunk,
//         link[c] = the run reaches the chunk's end AND the text goes on with the same byte
// pass 3 (kFinal = true):  RL[i] for every position, with carry[c + 1] = true run length at the first byte of the next chunk
// period p (1 for runs of one byte): position x is LINKED to x + 1 iff text[x] == text[x + p] (and x + p < n); a run is a
// maximal chain of links, its length in positions r, and what is stored is how far the text goes on repeating itself
// from there: RL = min(r - 1 + p, n - position).  For p = 1 that is the number of equal bytes, as before.
// (Round 5: a thread's 16 bytes and the 16 bytes one period further on are fetched as dwords and compared four at a time,
// the scan runs in registers, and the run lengths leave through LDS in coalesced lines: 1.14 -> ~0.3 ms for 128 MiB; the
// byte-wise loads, sixteen barriers and 64-byte-strided stores of the first version were 0.09 of the HBM peak.)
template <bool kFinal>
__global__ __launch_bounds__(kRunThreads) void runlen_chunk_kernel(const uint8_t *__restrict__ text, int64_t n,
                                                                   uint32_t *__restrict__ lead, uint8_t *__restrict__ link,
                                                                   const uint32_t *__restrict__ carry, uint32_t *__restrict__ RL,
                                                                   int period = 1)
{
    constexpr int kWaves = kRunThreads / kWave;
    __shared__ uint32_t w_a[kWaves];
    __shared__ uint8_t w_open[kWaves];
    __shared__ uint32_t s_first[kRunThreads + 1];                          // run length at every segment's first byte
    __shared__ uint32_t s_out[kFinal ? kRunThreads * (kRunPer + 1) : 1];   // (+1: padded rows, no bank conflicts)
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * kRunChunk;
    const int64_t end = c0 + kRunChunk < n ? c0 + kRunChunk : n;           // chunk = [c0, end)
    const int64_t p0 = c0 + (int64_t)t * kRunPer;
    // e bit i: position p0 + i is linked to the next one.  The text buffer is 16-byte aligned and followed by 64 zero
    // bytes and the rest of the workspace: the dwords below are always readable, and what lies at or beyond n is masked.
    uint32_t e = 0;
    if (p0 < end) {
        const uint4 va = *reinterpret_cast<const uint4 *>(text + p0);
        const uintptr_t qb = reinterpret_cast<uintptr_t>(text + p0 + period);
        const uint32_t *wq = reinterpret_cast<const uint32_t *>(qb & ~(uintptr_t)3);
        const uint32_t sh = (uint32_t)(qb & 3);
        const uint32_t q0 = wq[0], q1 = wq[1], q2 = wq[2], q3 = wq[3], q4 = wq[4];
        const uint32_t a4[4] = {va.x, va.y, va.z, va.w};
        const uint32_t b4[4] = {__builtin_amdgcn_alignbyte(q1, q0, sh), __builtin_amdgcn_alignbyte(q2, q1, sh),
                                __builtin_amdgcn_alignbyte(q3, q2, sh), __builtin_amdgcn_alignbyte(q4, q3, sh)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t x = a4[j] ^ b4[j];
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (((x >> (8 * b)) & 0xffu) == 0) e |= 1u << (4 * j + b);
        }
        // x + period < n
        const int64_t room = n - period - p0;                              // positions p0 + i with i < room may be linked
        if (room <= 0) e = 0;
        else if (room < kRunPer) e &= (1u << room) - 1u;
    }
    auto linked1 = [&](int64_t x) -> bool { return x + period < n && text[x] == text[x + period]; };
    const bool end_linked = end < n && linked1(end - 1);                   // the chunk's last position, to the next chunk
    // run lengths inside my segment, from the right; a position beyond `end` counts as a break
    uint32_t in[kRunPer];
    uint32_t run = 0;
#pragma unroll
    for (int i = kRunPer - 1; i >= 0; --i) {
        const bool valid = p0 + i < end;
        const bool cont = valid && i + 1 < kRunPer && p0 + i + 1 < end && ((e >> i) & 1u);
        run = valid ? (cont ? run + 1 : 1) : 0;
        in[i] = run;
    }
    // my segment's map: its leading run, open iff that run covers the whole segment and is linked to the next segment
    // (inside the chunk)
    const int64_t seg_end = p0 + kRunPer;
    const bool last_linked = ((e >> (kRunPer - 1)) & 1u) != 0;
    const bool full = p0 < end && in[0] == (uint32_t)kRunPer && seg_end < end && last_linked;
    const uint32_t first = affine_rscan_regs(p0 < end ? in[0] : 0u, full, w_a, w_open, t);
    // first = run length at my first byte, counted inside the chunk
    if (!kFinal) {
        if (t == 0) {
            lead[blockIdx.x] = first;
            link[blockIdx.x] = (c0 + (int64_t)first == end && end_linked) ? 1 : 0;
        }
        return;
    }
    s_first[t] = first;
    if (t == 0) s_first[kRunThreads] = 0;
    __syncthreads();
    const uint32_t next_seg = s_first[t + 1];                                // run length at the next segment's first byte
    const uint32_t next_chunk = end < n ? carry[blockIdx.x + 1] : 0;         // ... at the next chunk's first byte (true length)
#pragma unroll
    for (int i = 0; i < kRunPer; ++i) {
        const int64_t p = p0 + i;
        uint32_t r = in[i];
        // the run reaches my segment's end and goes on in the next segment (same chunk)?
        if (i + (int)r == kRunPer && seg_end < end && last_linked) r += next_seg;
        // ... reaches the chunk's end and goes on in the next chunk?
        if (p + (int64_t)r == end && end_linked) r += next_chunk;
        const int64_t far = (int64_t)r - 1 + period, left = n - p;
        s_out[t * (kRunPer + 1) + i] = (uint32_t)(far < left ? far : left);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kRunPer; ++i) {
        const int j = i * kRunThreads + t;                                   // position inside the chunk
        if (c0 + j < end) RL[c0 + j] = s_out[(j / kRunPer) * (kRunPer + 1) + (j % kRunPer)];
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
