// dq_radix.h -- radix constants shared by the digit-pass kernels (dq_onesweep.h).
//
// Keys are sorted 8 bits at a time, least significant digit first; one digit pass is ONE
// kernel (radix_rank_kernel, dq_onesweep.h).  HBM-bound integer work: no MFMA.
#pragma once
#include "dq_device_utils.h"

namespace dq {

constexpr int kRadixBits = 8;
constexpr int kRadixSize = 1 << kRadixBits;        // 256 digits == 256 threads

static_assert(kRadixSize == kBlock, "one thread per digit");

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift)
{
    return (uint32_t)(key >> shift) & (kRadixSize - 1);
}

}  // namespace dq
