// dq_radix.h -- stable LSD radix ranking of (64-bit key, suffix index) pairs.
//
// One digit pass = upsweep (per-workgroup digit histogram of a contiguous chunk)
//                + scan    (digit-major exclusive scan -> absolute output offsets)
//                + downsweep / radix_rank_scatter (re-read the chunk tile by tile,
//                  rank every key inside the tile with wave64 ballot multi-split
//                  and per-wave LDS histograms, stage the tile through LDS in
//                  digit order, write coalesced runs).
//
// HBM-bound integer work: no MFMA.  Algorithmic bytes per element per pass
// (K = 8 key bytes, w = sizeof(IdxT)): upsweep K read; downsweep 2*(K + w)
// (K + (K + w) when the index is synthesised in the first pass).
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kRadixBits = 8;
constexpr int kRadixSize = 1 << kRadixBits;        // 256 digits == 256 threads
constexpr int kItems = 16;                          // keys per thread per tile
constexpr int kTile = kBlock * kItems;              // 4096 keys per tile
constexpr int kWaveTile = kWave * kItems;           // 1024 keys per wave per tile
constexpr int kMaxSweepBlocks = 1024;               // 4 resident workgroups per CU

static_assert(kRadixSize == kBlock, "one thread per digit");

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift)
{
    return (uint32_t)(key >> shift) & (kRadixSize - 1);
}

// ---------------------------------------------------------------------------------
// upsweep: blockhist[g][d] = number of keys with digit d in workgroup g's chunk
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void radix_upsweep_kernel(
    const uint64_t *__restrict__ keys, int64_t m, int shift, int tiles_per_block,
    uint32_t *__restrict__ blockhist)
{
    __shared__ uint32_t hist[kWavesPerBlock][kRadixSize];
    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = lane_id();
    for (int i = tid; i < kWavesPerBlock * kRadixSize; i += kBlock) (&hist[0][0])[i] = 0;
    __syncthreads();

    const int64_t begin = (int64_t)blockIdx.x * tiles_per_block * kTile;
    int64_t end = begin + (int64_t)tiles_per_block * kTile;
    if (end > m) end = m;

    uint32_t *myhist = hist[w];
    int64_t base = begin;
    // full tiles: 16-byte loads (2 keys per lane), 8 loads in flight per lane
    for (; base + kTile <= end; base += kTile) {
        ulonglong2 v[kItems / 2];
#pragma unroll
        for (int j = 0; j < kItems / 2; ++j)
            v[j] = *reinterpret_cast<const ulonglong2 *>(keys + base + (int64_t)j * (2 * kBlock) + 2 * tid);
#pragma unroll
        for (int j = 0; j < kItems / 2; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t d = digit_of(h ? v[j].y : v[j].x, shift);
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                if (__all(d == d0)) {          // wave-uniform digit: one add instead of a 64-way conflict
                    if (lane == 0) myhist[d0] += kWave;
                } else {
                    atomicAdd(&myhist[d], 1u);
                }
            }
        }
    }
    // ragged tail
    for (int64_t i = base + tid; i < end; i += kBlock) atomicAdd(&myhist[digit_of(keys[i], shift)], 1u);
    __syncthreads();

    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) c += hist[i][tid];
    blockhist[(int64_t)blockIdx.x * kRadixSize + tid] = c;
}

// ---------------------------------------------------------------------------------
// scan: digit-major exclusive scan of blockhist[G][256] -> blockbase[G][256]
//       (absolute output offset of workgroup g's first key with digit d).
// One workgroup of 1024 threads = 256 digits x 4 segments of workgroups.
// ---------------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(1024) void radix_scan_kernel(const uint32_t *__restrict__ blockhist,
                                                           int nblocks, IdxT *__restrict__ blockbase)
{
    __shared__ IdxT seg_total[4][kRadixSize];
    __shared__ IdxT digit_start[kRadixSize];
    __shared__ IdxT wtmp[kWavesPerBlock];
    const int d = threadIdx.x & (kRadixSize - 1);
    const int seg = threadIdx.x >> 8;
    const int per = (nblocks + 3) / 4;
    const int g0 = seg * per;
    int g1 = g0 + per;
    if (g1 > nblocks) g1 = nblocks;

    IdxT sum = 0;
#pragma unroll 8
    for (int g = g0; g < g1; ++g) sum += (IdxT)blockhist[(int64_t)g * kRadixSize + d];
    seg_total[seg][d] = sum;
    __syncthreads();

    // exclusive scan over digits of the digit totals (threads 0..255)
    if (seg == 0) {
        IdxT tot = seg_total[0][d] + seg_total[1][d] + seg_total[2][d] + seg_total[3][d];
        const int l = lane_id();
        const int w = threadIdx.x >> 6;
        IdxT incl = wave_incl_sum(tot);
        if (l == kWave - 1) wtmp[w] = incl;
        digit_start[d] = incl - tot;           // wave-relative for now
    }
    __syncthreads();
    if (seg == 0) {
        const int w = threadIdx.x >> 6;
        IdxT off = 0;
        for (int i = 0; i < w; ++i) off += wtmp[i];
        digit_start[d] += off;
    }
    __syncthreads();

    IdxT run = digit_start[d];
    for (int s = 0; s < seg; ++s) run += seg_total[s][d];
#pragma unroll 4
    for (int g = g0; g < g1; ++g) {
        const IdxT c = (IdxT)blockhist[(int64_t)g * kRadixSize + d];
        blockbase[(int64_t)g * kRadixSize + d] = run;
        run += c;
    }
}

// ---------------------------------------------------------------------------------
// downsweep == radix_rank_scatter: the dominant kernel of the whole pipeline.
// ---------------------------------------------------------------------------------
template <typename IdxT, bool kSynthVals>
__global__ __launch_bounds__(kBlock) void radix_rank_scatter_kernel(
    const uint64_t *__restrict__ kin, const IdxT *__restrict__ vin,
    uint64_t *__restrict__ kout, IdxT *__restrict__ vout,
    int64_t m, int shift, int tiles_per_block, const IdxT *__restrict__ blockbase)
{
    __shared__ __attribute__((aligned(16))) uint64_t exch[kTile];      // 32 KiB, keys then values
    __shared__ uint32_t whist[kWavesPerBlock][kRadixSize];             // per-wave digit counters
    __shared__ uint32_t tile_base[kRadixSize];                         // tile-local digit starts
    __shared__ IdxT gofs[kRadixSize];                                  // global offset - tile start (may be < 0)
    __shared__ uint32_t wtmp[kWavesPerBlock];

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = lane_id();

    const int64_t tile0 = (int64_t)blockIdx.x * tiles_per_block;
    const int64_t ntiles_total = (m + kTile - 1) / kTile;
    int64_t tile1 = tile0 + tiles_per_block;
    if (tile1 > ntiles_total) tile1 = ntiles_total;
    if (tile0 >= tile1) return;

    IdxT gbase = blockbase[(int64_t)blockIdx.x * kRadixSize + tid];   // thread d owns digit d

    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t base = tile * kTile;
        const int valid = (m - base) < kTile ? (int)(m - base) : kTile;
        const int wbase = w * kWaveTile + lane;     // wave-striped: item k <-> element wbase + 64 k

        uint64_t key[kItems];
        IdxT val[kItems];
        if (valid == kTile) {
#pragma unroll
            for (int k = 0; k < kItems; ++k) key[k] = kin[base + wbase + k * kWave];
            if (!kSynthVals) {
#pragma unroll
                for (int k = 0; k < kItems; ++k) val[k] = vin[base + wbase + k * kWave];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kItems; ++k) {
                const int e = wbase + k * kWave;
                key[k] = e < valid ? kin[base + e] : ~0ull;     // padding sorts last, never written
            }
            if (!kSynthVals) {
#pragma unroll
                for (int k = 0; k < kItems; ++k) {
                    const int e = wbase + k * kWave;
                    val[k] = e < valid ? vin[base + e] : (IdxT)0;
                }
            }
        }
        if (kSynthVals) {
#pragma unroll
            for (int k = 0; k < kItems; ++k) val[k] = (IdxT)(base + wbase + k * kWave);
        }

        for (int i = tid; i < kWavesPerBlock * kRadixSize; i += kBlock) (&whist[0][0])[i] = 0;
        __syncthreads();

        // ---- rank inside the wave: ballot multi-split + running per-wave histogram ----
        uint32_t pos[kItems];
        uint32_t *myhist = whist[w];
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            const uint64_t peers = match_digit8(d);
            const uint32_t before = myhist[d];                 // same value for every peer
            const int r = mask_rank_lt(peers);
            if (r == 0) myhist[d] = before + (uint32_t)__popcll(peers);
            pos[k] = before + (uint32_t)r;
        }
        __syncthreads();

        // ---- per digit: exclusive prefix over waves, tile-level exclusive scan over digits ----
        {
            uint32_t c[kWavesPerBlock], tot = 0;
#pragma unroll
            for (int i = 0; i < kWavesPerBlock; ++i) { c[i] = whist[i][tid]; }
#pragma unroll
            for (int i = 0; i < kWavesPerBlock; ++i) { whist[i][tid] = tot; tot += c[i]; }
            uint32_t incl = wave_incl_sum(tot);
            if (lane == kWave - 1) wtmp[w] = incl;
            __syncthreads();
            uint32_t off = 0;
#pragma unroll
            for (int i = 0; i < kWavesPerBlock; ++i) if (i < w) off += wtmp[i];
            const uint32_t excl = off + incl - tot;
            tile_base[tid] = excl;
            gofs[tid] = gbase - (IdxT)excl;
            gbase += (IdxT)tot;       // (the padded tail only inflates digit 255 of the very last tile)
        }
        __syncthreads();

        // ---- stage keys in digit order through LDS, then write coalesced runs ----
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const uint32_t d = digit_of(key[k], shift);
            pos[k] += tile_base[d] + myhist[d];
            exch[pos[k]] = key[k];
        }
        __syncthreads();
        IdxT out[kItems];
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int q = k * kBlock + tid;
            const uint64_t kk = exch[q];
            out[k] = gofs[digit_of(kk, shift)] + (IdxT)q;
            if (q < valid) kout[out[k]] = kk;
        }
        __syncthreads();
        IdxT *exv = reinterpret_cast<IdxT *>(exch);
#pragma unroll
        for (int k = 0; k < kItems; ++k) exv[pos[k]] = val[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kItems; ++k) {
            const int q = k * kBlock + tid;
            if (q < valid) vout[out[k]] = exv[q];
        }
        __syncthreads();
    }
}

}  // namespace dq
