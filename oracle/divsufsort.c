/*
 * divsufsort.c -- instantiates the LibDivSufSort restatement (divsufsort_impl.h) for
 * 32- and 64-bit indices and keeps per-phase wall-clock timers.
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h): oracle + single-threaded CPU baseline.
 */
#define _POSIX_C_SOURCE 200809L
#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "dq_oracle.h"

#define ALPHABET_SIZE 256                                   /* DivSufSort.cs:14 */
#define BUCKET_A_SIZE ALPHABET_SIZE                         /* DivSufSort.cs:15 */
#define BUCKET_B_SIZE (ALPHABET_SIZE * ALPHABET_SIZE)       /* DivSufSort.cs:16 */
#define SS_BLOCKSIZE 1024                                   /* SsSort.cs:18 */
#define SS_INSERTIONSORT_THRESHOLD 8                        /* SsSort.cs:929 */
#define SS_STACK_SIZE 16                                    /* SsSort.cs:899 */
#define SS_MERGE_STACK_SIZE 32                              /* SsSort.cs:900 */
#define TR_INSERTIONSORT_THRESHOLD 8                        /* TrSort.cs:147 */
#define TR_STACK_SIZE 64                                    /* TrSort.cs:113 */

static _Thread_local double dss_phase[5];

static double dss_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void dq_oracle_last_phase_seconds(double out[5])
{
    for (int i = 0; i < 5; ++i) out[i] = dss_phase[i];
}

#define IDX int32_t
#define SUF _i32
#include "divsufsort_impl.h"
#undef IDX
#undef SUF

#define IDX int64_t
#define SUF _i64
#include "divsufsort_impl.h"
#undef IDX
#undef SUF
