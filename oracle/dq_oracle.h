/*
 * dq_oracle.h -- CPU oracle for the suffix-sorting hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in deltaq_amd/ (the product) links,
 * imports or executes this library; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may.  It restates, in plain C, the reference's
 * algorithm and the reference's own checkers for the path
 *   ISuffixSort.Sort  ->  LibDivSufSort.Sort  ->  DivSufSort.divsufsort
 * (reference: src/DeltaQ.SuffixSorting.LibDivSufSort/ (all .cs files),
 *  test/DeltaQ.SuffixSorting.LibDivSufSort.Tests/{LibDivSufSortTests,LDSSChecker}.cs).
 *
 * Parity pinning: the C# reference cannot be built or run in this image (no
 * dotnet/mono).  The oracle is pinned by (1) the uniqueness of the suffix
 * array under the reference's order, (2) the reference's own checkers restated
 * here (Verify, LDSSChecker.Check), (3) the reference's 13 binary fixtures and
 * seeded random buffers with golden digests under tests/golden/.
 */
#ifndef DQ_ORACLE_H
#define DQ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- result codes of the sufcheck restatement (LDSSChecker.cs:11-18) ---- */
#define DQ_CHECK_DONE            0
#define DQ_CHECK_BAD_ARGUMENTS  (-1)
#define DQ_CHECK_OUT_OF_RANGE   (-2)
#define DQ_CHECK_WRONG_ORDER    (-3)
#define DQ_CHECK_WRONG_POSITION (-4)

/* Restatement of LibDivSufSort (the reference's default ISuffixSort provider).
 * Returns 0 on success, -1 on bad arguments, -2 on allocation failure. */
int32_t dq_oracle_divsufsort_i32(const uint8_t *T, int32_t *SA, int64_t n);
int32_t dq_oracle_divsufsort_i64(const uint8_t *T, int64_t *SA, int64_t n);

/* Restatement of the reference's second provider, SAIS (src/DeltaQ.SuffixSorting.SAIS/SAIS.cs): induced
 * sorting, linear time.  Independent of the divsufsort restatement; the two must agree on every input.
 * Returns 0 on success, -1 on bad arguments, -2 on allocation failure. */
int32_t dq_oracle_sais_i32(const uint8_t *T, int32_t *SA, int64_t n);
int32_t dq_oracle_sais_i64(const uint8_t *T, int64_t *SA, int64_t n);

/* Per-phase wall time (seconds) of the last dq_oracle_divsufsort_* call on
 * this thread: [0] classify+bucket, [1] sssort, [2] trsort, [3] place B*,
 * [4] induce (construct_SA). */
void dq_oracle_last_phase_seconds(double out[5]);

/* Naive suffix array: comparison sort of suffixes with memcmp + length
 * tie-break (== ReadOnlySpan<byte>.SequenceCompareTo).  O(n^2 log n) worst
 * case; only for small n. */
int32_t dq_oracle_naive_sa_i32(const uint8_t *T, int32_t *SA, int64_t n);

/* Restatement of LibDivSufSortTests.Verify's strict-order loop
 * (LibDivSufSortTests.cs:43-59).  Returns -1 if every adjacent pair is
 * strictly increasing, else the first i with !(suffix(SA[i]) < suffix(SA[i+1])).
 * Values are NOT range-checked here (the reference would throw); call
 * sufcheck first. */
int64_t dq_oracle_verify_strict_i32(const uint8_t *T, const int32_t *SA, int64_t n);
int64_t dq_oracle_verify_strict_i64(const uint8_t *T, const int64_t *SA, int64_t n);

/* Strict-order check of `samples` pseudo-randomly chosen adjacent pairs
 * (splitmix64(seed)); for inputs too large for the full loop. Returns -1 or
 * the first failing i. */
int64_t dq_oracle_verify_sampled_i32(const uint8_t *T, const int32_t *SA, int64_t n,
                                     int64_t samples, uint64_t seed);
int64_t dq_oracle_verify_sampled_i64(const uint8_t *T, const int64_t *SA, int64_t n,
                                     int64_t samples, uint64_t seed);

/* Restatement of LDSSChecker.Check (LDSSChecker.cs:23-119), the port of
 * libdivsufsort's sufcheck.  sa_len is SA.Length (BadArguments when != n). */
int32_t dq_oracle_sufcheck_i32(const uint8_t *T, int64_t n, const int32_t *SA, int64_t sa_len);
int32_t dq_oracle_sufcheck_i64(const uint8_t *T, int64_t n, const int64_t *SA, int64_t sa_len);

/* The same check evaluated by `threads` threads (checkers_mt.c), for the full-size configurations:
 * same phases, same result codes. */
int32_t dq_oracle_sufcheck_mt_i32(const uint8_t *T, int64_t n, const int32_t *SA, int64_t sa_len, int32_t threads);
int32_t dq_oracle_sufcheck_mt_i64(const uint8_t *T, int64_t n, const int64_t *SA, int64_t sa_len, int32_t threads);

/* ---- the suffix array's consumer (bsdiff_scan.c): Diff.Create's match search and scan loop, restated ----
 * Search(I, old, new[scan..], 0, n, out pos) (Diff.cs:267-298) for scan = scans[q] (or scan0 + q when scans
 * is NULL), q < count.  SA has n entries; the zeroed sentinel I[n] (Diff.cs:78) is supplied internally. */
int32_t dq_oracle_bsdiff_search_i32(const uint8_t *old, int64_t n, const int32_t *SA, const uint8_t *nw, int64_t m,
                                    const int64_t *scans, int64_t scan0, int64_t count, int32_t *pos_out,
                                    int32_t *len_out);
int32_t dq_oracle_bsdiff_search_i64(const uint8_t *old, int64_t n, const int64_t *SA, const uint8_t *nw, int64_t m,
                                    const int64_t *scans, int64_t scan0, int64_t count, int64_t *pos_out,
                                    int64_t *len_out);
/* The scan loop of Diff.Create (Diff.cs:91-232) on raw streams: ctrl receives *nctrl (add, copy, seek)
 * triples (capacity 3 * (m + 1) int64), diff / extra the raw bytes (capacity m each). */
int32_t dq_oracle_bsdiff_scan_i32(const uint8_t *old, int64_t n, const int32_t *SA, const uint8_t *nw, int64_t m,
                                  int64_t *ctrl, int64_t *nctrl, uint8_t *diff, int64_t *ndiff, uint8_t *extra,
                                  int64_t *nextra, int64_t *searches);
int32_t dq_oracle_bsdiff_scan_i64(const uint8_t *old, int64_t n, const int64_t *SA, const uint8_t *nw, int64_t m,
                                  int64_t *ctrl, int64_t *nctrl, uint8_t *diff, int64_t *ndiff, uint8_t *extra,
                                  int64_t *nextra, int64_t *searches);
/* Patch.ApplyInternal (Patch.cs:95-168) on the same raw streams; 0 or -3 ("Corrupt patch"). */
int32_t dq_oracle_bspatch_apply(const uint8_t *old, int64_t n, const int64_t *ctrl, int64_t nctrl,
                                const uint8_t *diff, int64_t ndiff, const uint8_t *extra, int64_t nextra,
                                int64_t newsize, uint8_t *out);

/* .NET System.Random(int seed) compat generator (Knuth subtractive), used by
 * every reference test/bench buffer: new Random(670761).NextBytes(buf)
 * (LibDivSufSortTests.cs:29, SuffixSortingBenchmarks.cs:15). */
void dq_oracle_netrandom_bytes(int32_t seed, uint8_t *out, int64_t n);
int32_t dq_oracle_netrandom_first_sample(int32_t seed);

#ifdef __cplusplus
}
#endif
#endif
