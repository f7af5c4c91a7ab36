/*
 * checkers.c -- the reference's own acceptance checks, restated in C:
 *   Verify's strict-order loop   (LibDivSufSortTests.cs:43-59)
 *   LDSSChecker.Check / sufcheck (LDSSChecker.cs:23-119)
 * plus a naive suffix array used to pin everything on small inputs.
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h).
 */
#include <stdlib.h>
#include <string.h>
#include "dq_oracle.h"

#define IDX int32_t
#define SUF _i32
#include "checkers_impl.h"
#undef IDX
#undef SUF

#define IDX int64_t
#define SUF _i64
#include "checkers_impl.h"
#undef IDX
#undef SUF

/* ---- naive SA: sort suffix start positions by SequenceCompareTo order ---- */
static const uint8_t *g_T;
static int64_t g_n;

static int naive_cmp(const void *pa, const void *pb)
{
    int32_t a = *(const int32_t *)pa, b = *(const int32_t *)pb;
    return seqcmp_i32(g_T, g_n, a, b);
}

int32_t dq_oracle_naive_sa_i32(const uint8_t *T, int32_t *SA, int64_t n)
{
    if (n < 0 || n > 0x7fffffff) return -1;
    for (int64_t i = 0; i < n; ++i) SA[i] = (int32_t)i;
    g_T = T; g_n = n;               /* not re-entrant: tests call it serially */
    qsort(SA, (size_t)n, sizeof(int32_t), naive_cmp);
    return 0;
}
