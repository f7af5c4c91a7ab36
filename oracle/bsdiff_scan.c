/*
 * bsdiff_scan.c -- the consumer of the suffix array, restated: Diff.Create's match search and scan loop
 * (src/DeltaQ.BsDiff/Diff.cs) and Patch.ApplyInternal's reconstruction (src/DeltaQ.BsDiff/Patch.cs), on RAW
 * control / diff / extra streams (no BSDIFF40 container, no bzip2: SURVEY.md section 8(f) row 3 stays out).
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h): the checker for the HIP match search (dq_bsdiff_search_*).
 *
 *   dq_oracle_bsdiff_search_*   Search            Diff.cs:267-298  (+ MatchLength :248-265, CompareBytes :244-246)
 *   dq_oracle_bsdiff_scan_*     the scan loop     Diff.cs:91-232   -> (add, copy, seek) triples + diff + extra bytes
 *   dq_oracle_bspatch_apply     ApplyInternal     Patch.cs:95-168  -> the new file, from old + the three streams
 *
 * Pinning: the reference holds no golden patches; its own tests are round trips (BsDiffTests.cs:30-78:
 * Apply(old, Create(old, new)) == new on seeded buffers of 0 / 1 / 512 / 999 / 1024 / 4096 bytes).  The same
 * round trips run on this restatement in tests/test_bsdiff_oracle.py, and Search is cross-checked there
 * against a brute-force evaluation of its specification.
 */
#include <stdlib.h>
#include <string.h>
#include "dq_oracle.h"

/* ReadOnlySpan<byte>.SequenceCompareTo: unsigned bytes, a proper prefix sorts first  (Diff.cs:244-246) */
static int compare_bytes(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb)
{
    const int64_t m = la < lb ? la : lb;
    const int c = memcmp(a, b, (size_t)m);
    if (c != 0) return c;
    return (la > lb) - (la < lb);
}

/* Diff.cs:248-265 */
static int64_t match_length(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb)
{
    int64_t i;
    for (i = 0; i < la && i < lb; i++)
        if (a[i] != b[i]) break;
    return i;
}

#define DEFINE_SEARCH(IDX, SUF)                                                                              \
    /* Diff.cs:267-298.  I has n + 1 entries (Diff.cs:78: the last one is the zeroed sentinel slot). */       \
    static int64_t search##SUF(const IDX *I, const uint8_t *old, int64_t n, const uint8_t *nw, int64_t nwlen, \
                               int64_t start, int64_t end, int64_t *pos)                                      \
    {                                                                                                         \
        for (;;) {                                                                                            \
            if (end - start < 2) {                                                                            \
                const int64_t x = match_length(old + I[start], n - (int64_t)I[start], nw, nwlen);             \
                const int64_t y = match_length(old + I[end], n - (int64_t)I[end], nw, nwlen);                 \
                if (x > y) { *pos = (int64_t)I[start]; return x; }                                            \
                *pos = (int64_t)I[end];                                                                       \
                return y;                                                                                     \
            }                                                                                                 \
            const int64_t mid = start + (end - start) / 2;                                                    \
            if (compare_bytes(old + I[mid], n - (int64_t)I[mid], nw, nwlen) < 0) start = mid;                 \
            else end = mid;                                                                                   \
        }                                                                                                     \
    }                                                                                                         \
                                                                                                              \
    /* Search(I, old, new[scan..], 0, n, out pos) for scan = scan0 .. scan0 + count - 1 (Diff.cs:106).        \
     * SA has n entries; the sentinel I[n] = 0 is appended here. */                                           \
    int32_t dq_oracle_bsdiff_search##SUF(const uint8_t *old, int64_t n, const IDX *SA, const uint8_t *nw,     \
                                         int64_t m, const int64_t *scans, int64_t scan0, int64_t count,       \
                                         IDX *pos_out, IDX *len_out)                                          \
    {                                                                                                         \
        IDX *I = (IDX *)calloc((size_t)n + 1, sizeof(IDX));                                                   \
        if (!I) return -2;                                                                                    \
        if (n > 0) memcpy(I, SA, (size_t)n * sizeof(IDX));                                                    \
        for (int64_t q = 0; q < count; ++q) {                                                                 \
            const int64_t scan = scans ? scans[q] : scan0 + q;                                                \
            int64_t pos = 0;                                                                                  \
            const int64_t len = search##SUF(I, old, n, nw + scan, m - scan, 0, n, &pos);                      \
            pos_out[q] = (IDX)pos;                                                                            \
            len_out[q] = (IDX)len;                                                                            \
        }                                                                                                     \
        free(I);                                                                                              \
        return 0;                                                                                             \
    }                                                                                                         \
                                                                                                              \
    /* Diff.cs:91-232: the scan loop.  ctrl receives (add, copy, seek) triples (3 x int64 each), diff and     \
     * extra the raw byte streams; capacities: ctrl 3 * (m + 1) entries, diff m, extra m.                     \
     * searches (optional) counts the Search calls. */                                                        \
    int32_t dq_oracle_bsdiff_scan##SUF(const uint8_t *old, int64_t n, const IDX *SA, const uint8_t *nw,       \
                                       int64_t m, int64_t *ctrl, int64_t *nctrl, uint8_t *diff,               \
                                       int64_t *ndiff, uint8_t *extra, int64_t *nextra, int64_t *searches)    \
    {                                                                                                         \
        IDX *I = (IDX *)calloc((size_t)n + 1, sizeof(IDX));      /* :78 (n + 1), AllocationMode.Clear */       \
        if (!I) return -2;                                                                                    \
        if (n > 0) memcpy(I, SA, (size_t)n * sizeof(IDX));       /* :90 suffixSort.Sort(oldData, I[..^1]) */    \
        int64_t scan = 0, pos = 0, len = 0, lastscan = 0, lastpos = 0, lastoffset = 0;                        \
        int64_t nc = 0, nd = 0, ne = 0, ns = 0;                                                               \
        while (scan < m) {                                       /* :100 */                                   \
            int64_t oldscore = 0;                                                                             \
            int64_t scsc;                                                                                     \
            for (scsc = scan += len; scan < m; scan++) {         /* :104 */                                   \
                len = search##SUF(I, old, n, nw + scan, m - scan, 0, n, &pos);                                \
                ++ns;                                                                                         \
                for (; scsc < scan + len; scsc++)                                                             \
                    if ((scsc + lastoffset < n) && (old[scsc + lastoffset] == nw[scsc])) oldscore++;          \
                if ((len == oldscore && len != 0) || (len > oldscore + 8)) break;                             \
                if ((scan + lastoffset < n) && (old[scan + lastoffset] == nw[scan])) oldscore--;              \
            }                                                                                                 \
            if (len != oldscore || scan == m) {                  /* :127 */                                   \
                int64_t s = 0, sf = 0, lenf = 0;                                                              \
                for (int64_t i = 0; (lastscan + i < scan) && (lastpos + i < n);) {                            \
                    if (old[lastpos + i] == nw[lastscan + i]) s++;                                            \
                    i++;                                                                                      \
                    if (s * 2 - i > sf * 2 - lenf) { sf = s; lenf = i; }                                      \
                }                                                                                             \
                int64_t lenb = 0;                                                                             \
                if (scan < m) {                                  /* :147 */                                   \
                    s = 0;                                                                                    \
                    int64_t sb = 0;                                                                           \
                    for (int64_t i = 1; (scan >= lastscan + i) && (pos >= i); i++) {                          \
                        if (old[pos - i] == nw[scan - i]) s++;                                                \
                        if (s * 2 - i > sb * 2 - lenb) { sb = s; lenb = i; }                                  \
                    }                                                                                         \
                }                                                                                             \
                if (lastscan + lenf > scan - lenb) {             /* :167 */                                   \
                    const int64_t overlap = (lastscan + lenf) - (scan - lenb);                                \
                    s = 0;                                                                                    \
                    int64_t ss = 0, lens = 0;                                                                 \
                    for (int64_t i = 0; i < overlap; i++) {                                                   \
                        if (nw[lastscan + lenf - overlap + i] == old[lastpos + lenf - overlap + i]) s++;      \
                        if (nw[scan - lenb + i] == old[pos - lenb + i]) s--;                                  \
                        if (s > ss) { ss = s; lens = i + 1; }                                                 \
                    }                                                                                         \
                    lenf += lens - overlap;                                                                   \
                    lenb -= lens;                                                                             \
                }                                                                                             \
                for (int64_t i = 0; i < lenf; i++)               /* :196 diff string */                       \
                    diff[nd++] = (uint8_t)(nw[lastscan + i] - old[lastpos + i]);                              \
                const int64_t extra_len = (scan - lenb) - (lastscan + lenf);                                  \
                if (extra_len > 0) {                             /* :203 extra string */                      \
                    memcpy(extra + ne, nw + lastscan + lenf, (size_t)extra_len);                              \
                    ne += extra_len;                                                                          \
                }                                                                                             \
                ctrl[nc++] = lenf;                               /* :210-217 ctrl block */                    \
                ctrl[nc++] = extra_len;                                                                       \
                ctrl[nc++] = (pos - lenb) - (lastpos + lenf);                                                 \
                lastscan = scan - lenb;                                                                       \
                lastpos = pos - lenb;                                                                         \
                lastoffset = pos - scan;                                                                      \
            }                                                                                                 \
        }                                                                                                     \
        free(I);                                                                                              \
        *nctrl = nc / 3; *ndiff = nd; *nextra = ne;                                                           \
        if (searches) *searches = ns;                                                                         \
        return 0;                                                                                             \
    }

DEFINE_SEARCH(int32_t, _i32)
DEFINE_SEARCH(int64_t, _i64)

/* Patch.cs:95-168 on raw streams: for every triple, add `add` bytes of old to the diff string, copy `copy`
 * bytes of the extra string, seek `seek` in old.  Returns 0, or -3 for what the reference calls "Corrupt patch". */
int32_t dq_oracle_bspatch_apply(const uint8_t *old, int64_t n, const int64_t *ctrl, int64_t nctrl,
                                const uint8_t *diff, int64_t ndiff, const uint8_t *extra, int64_t nextra,
                                int64_t newsize, uint8_t *out)
{
    int64_t outpos = 0, oldpos = 0, dpos = 0, epos = 0;
    for (int64_t t = 0; outpos < newsize; ++t) {
        if (t >= nctrl) return -3;
        const int64_t add = ctrl[3 * t], copy = ctrl[3 * t + 1], seek = ctrl[3 * t + 2];
        /* (comparisons in a form that cannot overflow: the triples of a crafted patch are 63-bit values) */
        if (add < 0 || copy < 0 || add > newsize - outpos) return -3;
        /* short reads (:139-140): "Corrupt patch"; with add == 0 nothing is read wherever the position stands */
        if (add > ndiff - dpos || (add > 0 && (oldpos > n || add > n - oldpos))) return -3;
        for (int64_t i = 0; i < add; i++) out[outpos + i] = (uint8_t)(diff[dpos + i] + old[oldpos + i]);
        outpos += add; dpos += add; oldpos += add;
        if (copy > newsize - outpos || copy > nextra - epos) return -3;
        memcpy(out + outpos, extra + epos, (size_t)copy);
        outpos += copy; epos += copy;
        /* :165 Stream.Seek throws on a negative position */
        if (__builtin_add_overflow(oldpos, seek, &oldpos) || oldpos < 0) return -3;
    }
    return 0;
}
