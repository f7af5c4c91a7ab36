// dq_bucket_sort.h -- last step of the bucketed round 0: finish every bucket inside LDS.
//
// Random-like inputs sort their round-0 keys MSD-wise: two (three) digit passes of radix_rank_kernel
// order the packed words (key << ib | suffix) by the top 16 (24) key bits, i.e. into 65 536 (16 M)
// buckets of a few hundred to a few thousand words each; the remaining low key bits (up to 20) are
// then sorted bucket by bucket here, with no further global pass:
//
//   bucket_bounds_kernel   tile t nominally covers words [t*C, (t+1)*C); its real range starts at the
//                          first bucket boundary at or after t*C (binary search, one thread per tile),
//                          so a tile holds WHOLE buckets and at most C + X words (X = longest bucket
//                          the path accepts; a longer one raises the overflow flag and the host falls
//                          back to the plain digit passes)
//   bucket_sort_kernel     one workgroup per tile.  The words of a tile are already grouped by bucket,
//                          so inside the tile  key = (bucket - first bucket) << lowbits | low bits  is
//                          a small integer that is nearly uniform over its range.  It is sorted by
//                          direct placement:  bin = key * nbins / range  (monotone), bin counts by LDS
//                          atomics, scan, scatter of the 32-bit keys into bin order, then every element
//                          counts the members of its own bin that precede it (bins hold ~1 element).
//                          The suffix indices go through LDS into sorted order and leave as coalesced
//                          4/8-byte stores; "equals its predecessor" is known from the bin walk and
//                          becomes the tie bit (dq_ties.h: tie_collect_kernel reads those bits).
//                          Buckets never straddle tiles, so there are no seams to repair.
//
// Per suffix: 8 B read (word) + w B written (SA) + 1/8 B (tie bit): the cheapest pass of round 0,
// and it replaces two scattering passes (2 x 16 B) on 64 MiB ... 300 MiB inputs.  Pure streaming:
// no look-back, tiles are independent.
#pragma once
#include "dq_onesweep.h"

namespace dq {

constexpr int kBktCap = 12288;                             // words per tile at most (= threads x items of every geometry)
constexpr int kBktBins = kBktCap;                          // ~1 element per bin
constexpr int kBktMaxBin = 48;                             // a fuller bin = not the data this path is for

struct BucketFlags {
    // non-zero = the path gave up: 2 a bucket longer than X, 4 a tile spanning too many buckets, 8 a bin too
    // full (1 is tie_collect_kernel's "run of equal keys too long": the word is TieCounters::overflow)
    unsigned long long overflow;
};

// bounds[t] for t = 0..ntiles: first bucket boundary at or after t*C (bounds[ntiles] = n)
static __global__ __launch_bounds__(kBlock) void bucket_bounds_kernel(const uint64_t *__restrict__ W, int64_t n, int bshift,
                                                               int64_t C, int64_t X, int64_t ntiles,
                                                               int64_t *__restrict__ bounds,
                                                               BucketFlags *__restrict__ flags)
{
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t > ntiles) return;
    if (t == ntiles) { bounds[t] = n; return; }
    if (t == 0) { bounds[0] = 0; return; }
    const int64_t p = t * C;                                // p < n
    const uint64_t b = W[p - 1] >> bshift;
    int64_t lo = p, hi = p + X < n ? p + X : n;             // the boundary lies in [lo, hi] unless the bucket is too long
    if (hi < n && (W[hi] >> bshift) == b) {
        atomicOr(&flags->overflow, 2ull);                    // reason bits: see BucketFlags
        bounds[t] = p;
        return;
    }
    while (lo < hi) {                                       // first i with bucket(W[i]) != b; i == n counts
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((W[mid] >> bshift) == b) lo = mid + 1; else hi = mid;
    }
    bounds[t] = lo;
}

// bucket_sort_kernel: 1024 threads x 12 words; ONE persistent workgroup per CU (key buffer + bin table +
// suffixes = 144 KB of LDS) that walks over tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The words of the workgroup's
// NEXT tile are requested as soon as the current tile's words have left the load registers and stay in
// flight while the current tile is sorted (plain global loads: barriers do not drain them), so a tile does
// not wait for HBM -- as long as nothing in the loop spills: a scratch reload waits for every older load.
// Hence one register per element from load to store:  key << 6 | arrival number in the bin  -- unique inside
// the tile, so the place of an element in its bin is simply the number of smaller members; the suffixes wait
// in LDS.  What bounds the kernel (tools/kbench/bsort.hip stamps the phases): LDS accesses that hit random
// banks -- the bin count atomics, the bin starts, the walk over the bin and the two scatters cost ~7 cycles
// per wave access instead of 2 -- about 30 000 cycles per 12 288-word tile.
constexpr int kBktThreads = 1024;
constexpr int kBktItems = kBktCap / kBktThreads;
constexpr int kBktNB = 24576;                              // bins: load factor <= 1/2
constexpr int kBktArrBits = 6;                             // arrival numbers < 64 (kBktMaxBin = 48)
constexpr int kBktWalk = 4;                                // bin members inspected without a branch
static_assert(kBktMaxBin < (1 << kBktArrBits), "arrival numbers must fit");

// developer instrumentation (tools/kbench/bsort.hip): per-tile phase timestamps from thread 0
#ifdef DQ_BKT_PHASE_TIMING
__device__ long long *g_bkt_ts = nullptr;            // [ntiles][16]
#define DQ_BKT_PHASE(i) do { if (threadIdx.x == 0 && g_bkt_ts) g_bkt_ts[(long long)tile * 16 + (i)] = clock64(); } while (0)
#else
#define DQ_BKT_PHASE(i) do { } while (0)
#endif

// kExt: every word comes with one more byte of key (E[i] belongs to W[i]; radix_rank_kernel's kTextPackedExt / kKeysExt
// passes carried it along): it is appended to the word's low key bits, so the tile sorts by 8 more bits and the tie bits
// mean "equal in all of them".  The bytes of a tile are fetched when the tile starts (not a tile ahead like the words:
// the register file has no room for them), behind the zeroing of the bin table.
template <typename IdxT, bool kExt = false>
__global__ __launch_bounds__(kBktThreads, 4) void bucket_sort_kernel(
    const uint64_t *__restrict__ W, int ib, int lowbits, const int64_t *__restrict__ bounds, int64_t ntiles,
    IdxT *__restrict__ SA, uint32_t *__restrict__ ebits, BucketFlags *__restrict__ flags,
    const uint8_t *__restrict__ E = nullptr)
{
    constexpr int kBinsPerThread = kBktNB / kBktThreads;   // 24 sixteen-bit counters = 12 LDS words
    constexpr int kWords = kBinsPerThread / 2;
    __shared__ uint32_t buf[kBktCap + kBktWalk];            // unique keys in bin order, then suffixes in sorted order
    __shared__ uint32_t bins[kBktNB / 2 + 2];               // two 16-bit counters per word: counts, then starts
    __shared__ uint32_t sufs[kBktCap];                      // the tile's suffixes in load order (registers are scarce)
    __shared__ uint32_t wtot[kBktThreads / kWave];
    __shared__ uint64_t s_edge[2];                          // first and last word of the tile
    __shared__ uint32_t s_overflow;

    const uint32_t tid = threadIdx.x;
    const int lane = lane_id();
    const uint32_t wv = tid >> 6;
    const int bshift = ib + lowbits;
    const uint32_t imask = (uint32_t)((1ull << ib) - 1);
    const uint16_t *start16 = reinterpret_cast<const uint16_t *>(bins);

    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    // (lo, M) of the tile whose words are being fetched; M > capacity has been flagged by the bounds kernel
    int64_t lo_n = bounds[tile];
    uint32_t M_n = (uint32_t)(bounds[tile + 1] - lo_n);
    if (M_n > (uint32_t)kBktCap) M_n = 0;
    // ... and the bounds of the one after that: a dependent load in front of the fetch, so it is requested a
    // whole tile ahead of its use
    int64_t lo_nn = 0, hi_nn = 0;
    if (tile + gridDim.x < ntiles) { lo_nn = bounds[tile + gridDim.x]; hi_nn = bounds[tile + gridDim.x + 1]; }
    uint64_t wd[kBktItems];
    // Clamped, not predicated: the loads stay in flight together.  (And every array element is always assigned
    // unconditionally: a conditional element write turns the register array into one wide phi that the
    // allocator spills.)
#define DQ_BKT_FETCH()                                                            \
    do {                                                                          \
        const uint64_t *Wn = W + (M_n ? lo_n : 0);   /* (an empty tile may start at n) */ \
        const uint32_t last_ = M_n > 0 ? M_n - 1 : 0;                             \
        _Pragma("unroll") for (int k = 0; k < kBktItems; ++k) {                   \
            const uint32_t e_ = (uint32_t)(k * kBktThreads) + tid;                \
            wd[k] = Wn[e_ < last_ ? e_ : last_];                                  \
        }                                                                         \
    } while (0)
    DQ_BKT_FETCH();

    for (;;) {
        const int64_t lo = lo_n;
        const uint32_t M = M_n;
        DQ_BKT_PHASE(0);
        uint32_t ex[kExt ? kBktItems : 1];
        if (kExt) {
            const uint8_t *En = E + (M ? lo : 0);
            const uint32_t last_ = M > 0 ? M - 1 : 0;
#pragma unroll
            for (int k = 0; k < kBktItems; ++k) {
                const uint32_t e_ = (uint32_t)(k * kBktThreads) + tid;
                ex[k] = En[e_ < last_ ? e_ : last_];
            }
        }
        const int lowbits_t = kExt ? lowbits + 8 : lowbits;             // low key bits of the tile's keys
        // ---- this tile's words leave the fetch registers: key (relative to the tile's first bucket) and suffix ----
        if (tid == 0) { s_edge[0] = wd[0]; s_overflow = 0; }
        if (tid == kBktThreads - 1) s_edge[1] = wd[kBktItems - 1];          // (clamped loads: element M-1)
        for (uint32_t i = tid; i < (uint32_t)(kBktNB / 2 + 2); i += kBktThreads) bins[i] = 0;
        __syncthreads();
        DQ_BKT_PHASE(1);
        const uint64_t kfirst = s_edge[0] >> bshift, klast = s_edge[1] >> bshift;
        const uint64_t range = (klast - kfirst + 1) << lowbits_t;          // tile keys are < range
        // range << 6 must fit 32 bits
        const bool bad = M != 0 && range > (1ull << (32 - kBktArrBits));
        // bin = key * mult >> 32 < bins: mult <= bins * 2^32 / range.  (A float reciprocal is within 2^-22 of the
        // exact quotient; the factor 1 - 2^-20 keeps the product below it.  Any positive multiplier is monotone.)
        // (a tile of few, short buckets may have range <= bins: the multiplier then saturates just below 2^32
        // and bin = key - 1 or key, still monotone and < range <= bins)
        const uint32_t mult = (bad || M == 0) ? 0u
            : (uint32_t)fminf((float)kBktNB * 4294967296.0f * __frcp_rn((float)range) * (1.0f - 0x1p-20f), 4294967040.0f);
        const uint64_t kbase = kfirst << lowbits;
        uint32_t key[kBktItems];
#pragma unroll
        for (int k = 0; k < kBktItems; ++k) {
            key[k] = (uint32_t)((wd[k] >> ib) - kbase);
            if (kExt) key[k] = (key[k] << 8) | ex[k];
            sufs[k * kBktThreads + tid] = (uint32_t)wd[k] & imask;          // read back by this same thread
        }
        // ---- request the next tile ----
        const int64_t tn = tile + gridDim.x;
        if (tn < ntiles) {
            lo_n = lo_nn;
            M_n = (uint32_t)(hi_nn - lo_nn);
            if (M_n > (uint32_t)kBktCap) M_n = 0;
            DQ_BKT_FETCH();
            if (tn + gridDim.x < ntiles) { lo_nn = bounds[tn + gridDim.x]; hi_nn = bounds[tn + gridDim.x + 1]; }
        }
        if (bad) {
            if (tid == 0) atomicOr(&flags->overflow, 4ull);
        } else if (M != 0) {
            // ---- bin counts; the returned old count (arrival number) makes the key unique ----
#pragma unroll
            for (int k = 0; k < kBktItems; ++k) {
                const uint32_t e = (uint32_t)(k * kBktThreads) + tid;
                const uint32_t bin = __umulhi(key[k], mult);
                const uint32_t sh = (bin & 1u) * 16u;
                uint32_t old = 0;
                if (e < M) old = atomicAdd(&bins[bin >> 1], 1u << sh);
                key[k] = (key[k] << kBktArrBits) | ((old >> sh) & ((1u << kBktArrBits) - 1));
            }
            __syncthreads();
            DQ_BKT_PHASE(2);

            // ---- exclusive scan of the bin counts: thread t owns kBinsPerThread consecutive bins ----
            {
                uint32_t c[kWords];
                uint32_t sum = 0, mx = 0;
#pragma unroll
                for (int i = 0; i < kWords; ++i) {
                    c[i] = bins[tid * kWords + i];
                    const uint32_t a = c[i] & 0xffffu, b = c[i] >> 16;
                    sum += a + b;
                    mx = a > mx ? a : mx;
                    mx = b > mx ? b : mx;
                }
                if (mx > (uint32_t)kBktMaxBin) s_overflow = 1;
                const uint32_t incl = wave_incl_sum_dpp(sum);               // (DPP: no lane-index registers kept across the loop)
                if (lane == kWave - 1) wtot[wv] = incl;
                __syncthreads();
                uint32_t run = incl - sum;
#pragma unroll
                for (int i = 0; i < kBktThreads / kWave; ++i) if (i < (int)wv) run += wtot[i];
#pragma unroll
                for (int i = 0; i < kWords; ++i) {
                    const uint32_t a = c[i] & 0xffffu, b = c[i] >> 16;
                    bins[tid * kWords + i] = run | ((run + a) << 16);      // starts (<= 12288: 16 bits)
                    run += a + b;
                }
                if (tid == kBktThreads - 1) bins[kBktNB / 2] = run;         // start of the bin after the last: M
            }
            __syncthreads();
            DQ_BKT_PHASE(3);
            if (s_overflow) {                               // (uniform: written before the barrier above)
                if (tid == 0) atomicOr(&flags->overflow, 8ull);
            } else {
                // ---- scatter the unique keys into bin order ----
#pragma unroll
                for (int k = 0; k < kBktItems; ++k) {
                    const uint32_t e = (uint32_t)(k * kBktThreads) + tid;
                    const uint32_t s = start16[__umulhi(key[k] >> kBktArrBits, mult)] + (key[k] & ((1u << kBktArrBits) - 1));
                    if (e < M) buf[s] = key[k];
                }
                __syncthreads();
                DQ_BKT_PHASE(4);

                // ---- final place = bin start + smaller members of my bin; a member with my key and a smaller
                //      arrival number makes the tie bit of my final position.  The first kBktWalk members are
                //      read without a branch (a bin holds < 1/2 element on average); longer bins take the loop. ----
                uint32_t fin[kBktItems];
#pragma unroll
                for (int k = 0; k < kBktItems; ++k) {
                    const uint32_t e = (uint32_t)(k * kBktThreads) + tid;
                    uint32_t r = 0, tie = 0;
                    if (e < M) {                // (item slots beyond the tile's last element do no LDS work: a tile
                                                // of a 256 MiB text is 5/8 full, and whole items are then skipped)
                        const uint32_t bin = __umulhi(key[k] >> kBktArrBits, mult);
                        const uint32_t s0 = start16[bin], s1 = start16[bin + 1];
                        r = s0;
                        // (round 6: ~3/4 of the members are alone in their bin -- load factor 0.3 -- and have nothing to
                        // read; the others walk with a quarter of the lanes, i.e. with fewer bank conflicts)
                        if (s1 > s0 + 1) {
#pragma unroll
                        for (int q = 0; q < kBktWalk; ++q) {
                            const uint32_t o = buf[s0 + q];                 // (buf has kBktWalk entries of slack)
                            const uint32_t less = (s0 + q < s1) & (o < key[k]);
                            r += less;
                            tie |= less & ((o >> kBktArrBits) == (key[k] >> kBktArrBits));
                        }
                        }
                        if (s1 > s0 + kBktWalk) {
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
                            for (uint32_t j = s0 + kBktWalk; j < s1; ++j) {
                                const uint32_t o = buf[j];
                                const uint32_t less = o < key[k];
                                r += less;
                                tie |= less & ((o >> kBktArrBits) == (key[k] >> kBktArrBits));
                            }
                        }
                    }
                    fin[k] = e < M ? (r | (tie << 31)) : 0xffffffffu;      // final place (| tie flag), or "no element"
                }
                __syncthreads();                            // every key has been read: buf now takes the suffixes
                DQ_BKT_PHASE(5);
#pragma unroll
                for (int k = 0; k < kBktItems; ++k) {
                    if (fin[k] != 0xffffffffu)
                        buf[fin[k] & 0x7fffffffu] = sufs[k * kBktThreads + tid] | (fin[k] & 0x80000000u);
                }
                __syncthreads();
                DQ_BKT_PHASE(6);
                // ---- sorted suffixes out, coalesced; the few tie bits by atomic OR ----
                IdxT *out = SA + lo;
                uint32_t v[kBktItems];
#pragma unroll
                for (int k = 0; k < kBktItems; ++k) {
                    const uint32_t e = (uint32_t)(k * kBktThreads) + tid;
                    v[k] = buf[e < M ? e : 0];
                }
                uint32_t ties = 0;
#pragma unroll
                for (int k = 0; k < kBktItems; ++k) {
                    const uint32_t e = (uint32_t)(k * kBktThreads) + tid;
                    if (e < M) out[e] = (IdxT)(v[k] & 0x7fffffffu);         // (n <= 2^31 on this path: a suffix fits 31 bits)
                    ties |= (e < M && (v[k] >> 31)) ? (1u << k) : 0u;
                }
                while (ties) {
                    const int k = __builtin_ctz(ties);
                    ties &= ties - 1;
                    const uint64_t o = (uint64_t)(lo + k * kBktThreads + tid);
                    atomicOr(&ebits[o >> 5], 1u << ((uint32_t)o & 31u));
                }
                DQ_BKT_PHASE(7);
            }
        }
        if (tn >= ntiles) break;
        tile = tn;
        __syncthreads();                                    // buf, bins and s_edge are reused
    }
#undef DQ_BKT_FETCH
}

}  // namespace dq
