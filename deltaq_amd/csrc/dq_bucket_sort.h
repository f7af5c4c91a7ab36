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
    unsigned long long overflow;        // a bucket longer than X, a tile spanning too many buckets, a bin too full
};

// bounds[t] for t = 0..ntiles: first bucket boundary at or after t*C (bounds[ntiles] = n)
__global__ __launch_bounds__(kBlock) void bucket_bounds_kernel(const uint64_t *__restrict__ W, int64_t n, int bshift,
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
        atomicExch(&flags->overflow, 1ull);
        bounds[t] = p;
        return;
    }
    while (lo < hi) {                                       // first i with bucket(W[i]) != b; i == n counts
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((W[mid] >> bshift) == b) lo = mid + 1; else hi = mid;
    }
    bounds[t] = lo;
}

// Geometry: kBktThreads x kBktItems = kBktCap; kMinWaves waves per SIMD are asked for (two workgroups per CU:
// 73.8 KB of LDS each).
template <typename IdxT, int kBktThreads, int kBktItems, int kMinWaves>
__global__ __launch_bounds__(kBktThreads, kMinWaves) void bucket_sort_kernel(
    const uint64_t *__restrict__ W, int ib, int lowbits, const int64_t *__restrict__ bounds,
    IdxT *__restrict__ SA, uint32_t *__restrict__ ebits, BucketFlags *__restrict__ flags)
{
    static_assert(kBktThreads * kBktItems == kBktCap, "one tile per workgroup");
    constexpr int kBktBinsPerThread = kBktBins / kBktThreads;
    __shared__ uint32_t buf[kBktCap];                       // keys in bin order, then suffixes in sorted order
    __shared__ uint32_t bins[kBktBins / 2];                 // two 16-bit counters per word: counts, then starts
    __shared__ uint32_t wtot[kBktThreads / kWave];
    __shared__ uint32_t s_overflow;

    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wv = tid >> 6;
    const int64_t lo = bounds[blockIdx.x];
    const int M = (int)(bounds[blockIdx.x + 1] - lo);
    if (M <= 0) return;
    if (M > kBktCap) { if (tid == 0) atomicExch(&flags->overflow, 1ull); return; }   // (bounds kernel flagged it too)

    // ---- the tile's key range (uniform over the workgroup) ----
    const int bshift = ib + lowbits;
    const uint64_t kfirst = W[lo] >> bshift, klast = W[lo + M - 1] >> bshift;
    const uint64_t range = (klast - kfirst + 1) << lowbits;                // tile keys are < range
    // (range > kBktBins keeps the multiplier below 2^32; lowbits >= 14 guarantees it)
    if (klast - kfirst >= 4096 || range > (1ull << 32) || range <= (uint64_t)kBktBins) {
        if (tid == 0) atomicExch(&flags->overflow, 1ull);
        return;
    }
    const uint32_t mult = (uint32_t)(((uint64_t)kBktBins << 32) / range);  // bin = key * mult >> 32 < kBktBins
    const uint64_t kbase = kfirst << lowbits;
    const uint32_t imask = (uint32_t)((1ull << ib) - 1);

    for (int i = tid; i < kBktBins / 2; i += kBktThreads) bins[i] = 0;
    if (tid == 0) s_overflow = 0;

    // ---- load; element e = k * kBktThreads + tid of the tile: tile-relative key and suffix, 32 bits each ----
    uint32_t key[kBktItems], idx[kBktItems];
#pragma unroll
    for (int k = 0; k < kBktItems; ++k) {
        const int e = k * kBktThreads + tid;
        const uint64_t wd = W[lo + (e < M ? e : M - 1)];          // (clamped, not predicated: the loads stay in flight together)
        key[k] = (uint32_t)((wd >> ib) - kbase);
        idx[k] = (uint32_t)wd & imask;
    }
    __syncthreads();

    // ---- bin counts; the returned old count is the element's arrival number inside its bin ----
    uint32_t slot[kBktItems];
#pragma unroll
    for (int k = 0; k < kBktItems; ++k) {
        // (every array element is assigned unconditionally: a conditional element write turns the whole
        // register array into one wide phi and the allocator spills it)
        const int e = k * kBktThreads + tid;
        const uint32_t bin = __umulhi(key[k], mult);
        const uint32_t sh = (bin & 1u) * 16u;
        uint32_t old = 0;
        if (e < M) old = atomicAdd(&bins[bin >> 1], 1u << sh);
        slot[k] = (old >> sh) & 0xffffu;
    }
    __syncthreads();

    // ---- exclusive scan of the bin counts: thread t owns kBktBinsPerThread consecutive bins ----
    {
        constexpr int kWords = kBktBinsPerThread / 2;
        uint32_t sum = 0, mx = 0;
#pragma unroll
        for (int i = 0; i < kWords; ++i) {
            const uint32_t c = bins[tid * kWords + i];
            const uint32_t a = c & 0xffffu, b = c >> 16;
            sum += a + b;
            mx = a > mx ? a : mx;
            mx = b > mx ? b : mx;
        }
        if (mx > (uint32_t)kBktMaxBin) s_overflow = 1;
        const uint32_t incl = wave_incl_sum(sum);
        if (lane == kWave - 1) wtot[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
#pragma unroll
        for (int i = 0; i < kBktThreads / kWave; ++i) if (i < wv) run += wtot[i];
#pragma unroll
        for (int i = 0; i < kWords; ++i) {
            const uint32_t c = bins[tid * kWords + i];
            const uint32_t a = c & 0xffffu, b = c >> 16;
            bins[tid * kWords + i] = run | ((run + a) << 16);      // starts (<= 12288: 16 bits)
            run += a + b;
        }
    }
    __syncthreads();
    if (s_overflow) { if (tid == 0) atomicExch(&flags->overflow, 1ull); return; }

    // ---- scatter the 32-bit keys into bin order ----
    const uint16_t *start16 = reinterpret_cast<const uint16_t *>(bins);     // start of bin b (b < kBktBins)
#pragma unroll
    for (int k = 0; k < kBktItems; ++k) {
        const int e = k * kBktThreads + tid;
        const uint32_t s = slot[k] + start16[__umulhi(key[k], mult)];
        slot[k] = s;
        if (e < M) buf[s] = key[k];
    }
    __syncthreads();

    // ---- final place = bin start + members of my bin that precede me (smaller key, or equal key and
    //      earlier slot); "an equal key precedes me" is the tie bit of my final position ----
#pragma unroll
    for (int k = 0; k < kBktItems; ++k) {
        const int e = k * kBktThreads + tid;
        uint32_t f = 0xffffffffu;
        if (e < M) {
            const uint32_t bin = __umulhi(key[k], mult);
            const uint32_t s0 = start16[bin];
            const uint32_t s1 = bin + 1 < (uint32_t)kBktBins ? (uint32_t)start16[bin + 1] : (uint32_t)M;
            uint32_t r = s0, tie = 0;
#pragma unroll 1
            for (uint32_t j = s0; j < s1; ++j) {
                const uint32_t o = buf[j];
                const uint32_t eq_before = (o == key[k]) & (j < slot[k]);
                r += (o < key[k]) | eq_before;
                tie |= eq_before;
            }
            f = r | (tie << 31);
        }
        slot[k] = f;                                        // final place (| tie flag), or ~0 for "no element"
    }
    __syncthreads();                                        // every key has been read: buf now takes the suffixes
#pragma unroll
    for (int k = 0; k < kBktItems; ++k) {
        if (slot[k] != 0xffffffffu) buf[slot[k] & 0x7fffffffu] = idx[k] | (slot[k] & 0x80000000u);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kBktItems; ++k) {
        const int e = k * kBktThreads + tid;
        if (e < M) {
            const uint32_t v = buf[e];
            const int64_t o = lo + e;
            SA[o] = (IdxT)(v & 0x7fffffffu);                 // (n <= 2^31 on this path: a suffix fits 31 bits)
            if (v >> 31) atomicOr(&ebits[(uint64_t)o >> 5], 1u << ((uint32_t)o & 31u));
        }
    }
}

}  // namespace dq
