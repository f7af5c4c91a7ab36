// dq_device_utils.h -- wave64 / workgroup primitives for gfx950 (CDNA4).
// Everything here assumes 64-lane wavefronts and 256-thread workgroups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dq {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

// The bound every look-back spin gives up at (radix_rank_kernel, seg_fused_kernel): kSpinLimit polls -- seconds -- in
// production.  It is a KERNEL ARGUMENT (spin_limit, last parameter of both kernels): the fault-injection tests
// (DQ_FAULT=spin, dq_runtime.h) launch with 0, so that the first empty poll raises the error word and the host's error
// path runs -- per launch of the calling thread, nothing device-wide that a concurrent caller could see or reset
// (round-5 advice: it was a `static __device__` word per translation unit, rewritten through hipMemcpyToSymbol).
constexpr uint32_t kSpinLimit = 1u << 24;

__device__ __forceinline__ int lane_id()
{
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `mask` at lane positions strictly below the calling lane
__device__ __forceinline__ int mask_rank_lt(uint64_t mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Lanes of the wave whose 8-bit digit equals this lane's digit (wave64 multi-split
// by ballot: 8 ballots, one per digit bit).
__device__ __forceinline__ uint64_t match_digit8(uint32_t d)
{
    uint64_t mask = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
    }
    return mask;
}

template <typename T>
__device__ __forceinline__ T wave_incl_sum(T v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        T t = __shfl_up(v, o, kWave);
        if (l >= o) v += t;
    }
    return v;
}

// Inclusive prefix sum over the 64 lanes with DPP only (no lane-index registers, no LDS permute):
// Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8, zeros shifted in), then lane 15 of rows
// 0 / 2 added to rows 1 / 3 (row_bcast:15) and lane 31 to rows 2 and 3 (row_bcast:31).
__device__ __forceinline__ uint32_t wave_incl_sum_dpp(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_incl_max(T v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        T t = __shfl_up(v, o, kWave);
        if (l >= o) v = t > v ? t : v;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, kWave);
        v = t > v ? t : v;
    }
    return v;
}

// Exclusive prefix sum over the 256 threads of a workgroup; `tmp` is kWavesPerBlock
// words of LDS.  Returns the exclusive prefix; *total receives the block sum.
// Contains two __syncthreads(); tmp may be reused after it returns only after
// another barrier.
template <typename T>
__device__ __forceinline__ T block_excl_sum(T v, T *tmp, T *total)
{
    const int l = lane_id();
    const int w = threadIdx.x >> 6;
    T incl = wave_incl_sum(v);
    if (l == kWave - 1) tmp[w] = incl;
    __syncthreads();
    T off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) {
        T t = tmp[i];
        if (i < w) off += t;
        tot += t;
    }
    *total = tot;
    return off + incl - v;
}

// Exclusive prefix max over the 256 threads of a workgroup; identity is -1 (the
// values are non-negative indices).  One __syncthreads(); `tmp` as above.
template <typename T>
__device__ __forceinline__ T block_excl_max(T v, T *tmp)
{
    const int l = lane_id();
    const int w = threadIdx.x >> 6;
    const T incl = wave_incl_max(v);
    if (l == kWave - 1) tmp[w] = incl;
    T excl = __shfl_up(incl, 1, kWave);
    if (l == 0) excl = (T)-1;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kWavesPerBlock; ++i) {
        const T t = tmp[i];
        if (i < w) excl = t > excl ? t : excl;
    }
    return excl;
}

}  // namespace dq
