/*
 * sais.c -- restatement of the reference's SAIS provider (src/DeltaQ.SuffixSorting.SAIS/SAIS.cs), the second,
 * linear-time CPU implementation the oracle carries beside the LibDivSufSort restatement (divsufsort.c).
 * Two independent algorithms that must agree on every input; the full-size configurations are bit-compared
 * against both (SAIS is linear-time whatever the input's repetitiveness; on this host it runs at about half the
 * divsufsort restatement's speed on random and text-like data).  TEST INFRASTRUCTURE ONLY (see dq_oracle.h).
 */
#include <stdlib.h>
#include <string.h>
#include "dq_oracle.h"

/* ---- 32-bit indices ---- */
#define IDX int32_t
#define REC _int_i32
#define CHR int32_t
#define LVL _int_i32
#include "sais_impl.h"
#undef CHR
#undef LVL
#define CHR uint8_t
#define LVL _u8_i32
#include "sais_impl.h"
#undef CHR
#undef LVL
#undef REC
#undef IDX

/* ---- 64-bit indices (inputs beyond the reference's int interface) ---- */
#define IDX int64_t
#define REC _int_i64
#define CHR int64_t
#define LVL _int_i64
#include "sais_impl.h"
#undef CHR
#undef LVL
#define CHR uint8_t
#define LVL _u8_i64
#include "sais_impl.h"
#undef CHR
#undef LVL
#undef REC
#undef IDX

/* SAIS.Sort(textBuffer, suffixBuffer), SAIS.cs:25-41: n <= 1 handled here, else sais_main(T, SA, 0, n, 256). */
int32_t dq_oracle_sais_i32(const uint8_t *T, int32_t *SA, int64_t n)
{
    if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!T || !SA))) return -1;
    if (n <= 1) { if (n == 1) SA[0] = 0; return 0; }
    return sais_main_u8_i32(T, SA, 0, (int32_t)n, 256);
}

int32_t dq_oracle_sais_i64(const uint8_t *T, int64_t *SA, int64_t n)
{
    if (n < 0 || (n > 0 && (!T || !SA))) return -1;
    if (n <= 1) { if (n == 1) SA[0] = 0; return 0; }
    return sais_main_u8_i64(T, SA, 0, n, 256);
}
