// dq_coded_keys.h -- 64-bit round-0 keys from alphabetic codewords (see dq_alpha_code.h for the code).
//
// key(i) = the first 64 bits of  code(T[i]) code(T[i+1]) ...  (zeros behind the end of the text).  A lane owns 4
// consecutive suffixes i0 .. i0+3 (i0 a multiple of 4) and reads the 20 bytes T[i0 .. i0+19] as 5 dwords: the
// codewords are appended once into a 128-bit accumulator and the 4 keys are 64-bit windows of it, at the bit
// offsets where the codewords of T[i0+1], T[i0+2], T[i0+3] begin.  Every codeword has 4...8 bits, so 16
// characters always fill a window: the last window needs T[i0+3 .. i0+18], and appending stops as soon as it
// is complete (at most 24 + 64 + 7 bits: the accumulator cannot overflow).
//
// Plain C++ so that the CPU tests compile the very same function (tests/native/alpha_harness.cpp).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DQ_HD __host__ __device__
#else
#define DQ_HD
#endif

namespace dq {

// tab[b] = codeword << 4 | length (dq_alpha_code.h); w[0..4] = the 5 little-endian dwords at T[i0]
DQ_HD inline void coded_keys4(const uint32_t w[5], const uint16_t *tab, uint64_t key[4])
{
    // 128-bit accumulator as two words; shifts by 0...8 (12) and 0...47 bits written so that a count of 0 is fine:
    // x >> (64 - s) == (x >> 1) >> (63 - s) for s >= 1, and 0 for s = 0
    uint64_t hi = 0, lo = 0;
    uint32_t bits = 0, need = 0xffffffffu, off[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 19; ++i) {
        const uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
        const uint32_t e = tab[b];
        if (i >= 1 && i <= 3) off[i] = bits;
        if (i == 3) need = bits + 64;
        const bool take = bits < need;
        const uint32_t len = take ? (e & 15u) : 0u;
        const uint32_t code = take ? (e >> 4) : 0u;
        hi = (hi << len) | ((lo >> 1) >> (63u - len));
        lo = (lo << len) | code;
        bits += len;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t s = bits - off[c] - 64;          // 0 ... 47: the window's distance from the accumulator's low end
        key[c] = ((hi << 1) << (63u - s)) | (lo >> s);
    }
}

}  // namespace dq
