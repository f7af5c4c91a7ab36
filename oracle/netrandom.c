/*
 * netrandom.c -- .NET System.Random(int Seed) (the seeded "compat"
 * implementation: Knuth's subtractive generator), needed to regenerate the
 * buffers the reference's tests and benchmark use:
 *   new Random(63*13*63*13).NextBytes(buf)   LibDivSufSortTests.cs:29-41
 *   new Random(63*13*63*13)                  SuffixSortingBenchmarks.cs:15
 * Check values: Random(0).Next() == 1559595546, Random(42).Next() == 1434747710.
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h).
 */
#include "dq_oracle.h"

#define MBIG  2147483647
#define MSEED 161803398

typedef struct { int32_t seed_array[56]; int inext, inextp; } netrandom_t;

static void netrandom_init(netrandom_t *r, int32_t seed)
{
    int32_t subtraction = (seed == INT32_MIN) ? INT32_MAX : (seed < 0 ? -seed : seed);
    int32_t mj = MSEED - subtraction;
    int32_t mk = 1;
    for (int i = 0; i < 56; ++i) r->seed_array[i] = 0;
    r->seed_array[55] = mj;
    for (int i = 1; i < 55; ++i) {
        int ii = (21 * i) % 55;
        r->seed_array[ii] = mk;
        mk = mj - mk;
        if (mk < 0) mk += MBIG;
        mj = r->seed_array[ii];
    }
    for (int k = 1; k < 5; ++k) {
        for (int i = 1; i < 56; ++i) {
            /* int32 wrap-around subtraction, as in C# unchecked arithmetic */
            uint32_t d = (uint32_t)r->seed_array[i] - (uint32_t)r->seed_array[1 + (i + 30) % 55];
            r->seed_array[i] = (int32_t)d;
            if (r->seed_array[i] < 0) r->seed_array[i] += MBIG;
        }
    }
    r->inext = 0;
    r->inextp = 21;
}

static int32_t netrandom_sample(netrandom_t *r)
{
    int loc_inext = r->inext, loc_inextp = r->inextp;
    if (++loc_inext >= 56) loc_inext = 1;
    if (++loc_inextp >= 56) loc_inextp = 1;
    int32_t ret = (int32_t)((uint32_t)r->seed_array[loc_inext] - (uint32_t)r->seed_array[loc_inextp]);
    if (ret == MBIG) ret--;
    if (ret < 0) ret += MBIG;
    r->seed_array[loc_inext] = ret;
    r->inext = loc_inext;
    r->inextp = loc_inextp;
    return ret;
}

void dq_oracle_netrandom_bytes(int32_t seed, uint8_t *out, int64_t n)
{
    netrandom_t r;
    netrandom_init(&r, seed);
    /* NextBytes: buffer[i] = (byte)InternalSample() */
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)netrandom_sample(&r);
}

int32_t dq_oracle_netrandom_first_sample(int32_t seed)
{
    netrandom_t r;
    netrandom_init(&r, seed);
    return netrandom_sample(&r);
}
