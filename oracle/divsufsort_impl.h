/*
 * divsufsort_impl.h -- CPU restatement of the reference's managed LibDivSufSort,
 * instantiated for int32_t and int64_t indices by divsufsort.c (define IDX, SUF).
 *
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h): the oracle the HIP path is compared
 * with, and bench.py's single-threaded cpu_baseline ("port").  Never shipped, never
 * called by deltaq_amd/.
 *
 * Every function cites the reference code it follows
 * (src/DeltaQ.SuffixSorting.LibDivSufSort/...).  The reference indexes one Span<int>
 * with integer offsets (SAPtr); here the same offsets are C pointers into SA.
 * Constants: SsSort.cs:18,899-900,929; TrSort.cs:113,147; DivSufSort.cs:14-16.
 */

#define DSS_CAT_(a, b) a##b
#define DSS_CAT(a, b) DSS_CAT_(a, b)
#define FN(name) DSS_CAT(name, SUF)

#define SWAP_IDX(a, b) do { IDX t__ = (a); (a) = (b); (b) = t__; } while (0)
#define SWAP_PTR(a, b) do { IDX *t__ = (a); (a) = (b); (b) = t__; } while (0)

/* Utils.cs:29-31,56,90: tr_ilg / ss_ilg == BitOperations.Log2((uint)n); 0 for n == 0 */
static inline int FN(ilg)(IDX n)
{
    return n > 0 ? 63 - __builtin_clzll((unsigned long long)n) : 0;
}

/* SsSort.cs:1573,1634-1642: min(SS_BLOCKSIZE, (int)sqrt(x)) */
static inline IDX FN(ss_isqrt)(IDX x)
{
    if (x >= (IDX)SS_BLOCKSIZE * SS_BLOCKSIZE) return SS_BLOCKSIZE;
    return (IDX)sqrtf((float)x);
}

/* ------------------------------------------------------------------ sssort */

/* SsSort.cs:155-194 ss_compare: compares T[p1[0]+depth .. p1[1]+2) with the same of p2 */
static inline int FN(ss_compare)(const uint8_t *T, const IDX *p1, const IDX *p2, IDX depth)
{
    const uint8_t *U1 = T + depth + p1[0], *U2 = T + depth + p2[0];
    const uint8_t *U1n = T + p1[1] + 2, *U2n = T + p2[1] + 2;
    while (U1 < U1n && U2 < U2n && *U1 == *U2) { ++U1; ++U2; }
    return U1 < U1n ? (U2 < U2n ? (int)*U1 - (int)*U2 : 1) : (U2 < U2n ? -1 : 0);
}

/* SsSort.cs:1434-1489 ss_insertionsort */
static void FN(ss_insertionsort)(const uint8_t *T, const IDX *PA, IDX *first, IDX *last, IDX depth)
{
    IDX *i, *j;
    IDX t;
    int r;
    for (i = last - 2; first <= i; --i) {
        for (t = *i, j = i + 1; 0 < (r = FN(ss_compare)(T, PA + t, PA + *j, depth));) {
            do { *(j - 1) = *j; } while ((++j < last) && (*j < 0));
            if (last <= j) break;
        }
        if (r == 0) *j = ~*j;
        *(j - 1) = t;
    }
}

/* SsSort.cs:1529-1567 ss_fixdown */
static inline void FN(ss_fixdown)(const uint8_t *Td, const IDX *PA, IDX *SA, IDX i, IDX size)
{
    IDX j, k, v;
    int c, d, e;
    for (v = SA[i], c = Td[PA[v]]; (j = 2 * i + 1) < size; SA[i] = SA[k], i = k) {
        d = Td[PA[SA[k = j++]]];
        if (d < (e = Td[PA[SA[j]]])) { k = j; d = e; }
        if (d <= c) break;
    }
    SA[i] = v;
}

/* SsSort.cs:1492-1527 ss_heapsort */
static void FN(ss_heapsort)(const uint8_t *Td, const IDX *PA, IDX *SA, IDX size)
{
    IDX i, m, t;
    m = size;
    if ((size % 2) == 0) {
        m--;
        if (Td[PA[SA[m / 2]]] < Td[PA[SA[m]]]) SWAP_IDX(SA[m], SA[m / 2]);
    }
    for (i = m / 2 - 1; 0 <= i; --i) FN(ss_fixdown)(Td, PA, SA, i, m);
    if ((size % 2) == 0) { SWAP_IDX(SA[0], SA[m]); FN(ss_fixdown)(Td, PA, SA, 0, m); }
    for (i = m - 1; 0 < i; --i) {
        t = SA[0]; SA[0] = SA[i];
        FN(ss_fixdown)(Td, PA, SA, 0, i);
        SA[i] = t;
    }
}

/* SsSort.cs:1348-1370 ss_median3 */
static inline IDX *FN(ss_median3)(const uint8_t *Td, const IDX *PA, IDX *v1, IDX *v2, IDX *v3)
{
    if (Td[PA[*v1]] > Td[PA[*v2]]) SWAP_PTR(v1, v2);
    if (Td[PA[*v2]] > Td[PA[*v3]]) return Td[PA[*v1]] > Td[PA[*v3]] ? v1 : v3;
    return v2;
}

/* SsSort.cs:1310-1342 ss_median5 */
static inline IDX *FN(ss_median5)(const uint8_t *Td, const IDX *PA, IDX *v1, IDX *v2, IDX *v3,
                                  IDX *v4, IDX *v5)
{
    if (Td[PA[*v2]] > Td[PA[*v3]]) SWAP_PTR(v2, v3);
    if (Td[PA[*v4]] > Td[PA[*v5]]) SWAP_PTR(v4, v5);
    if (Td[PA[*v2]] > Td[PA[*v4]]) { SWAP_PTR(v2, v4); SWAP_PTR(v3, v5); }
    if (Td[PA[*v1]] > Td[PA[*v3]]) SWAP_PTR(v1, v3);
    if (Td[PA[*v1]] > Td[PA[*v4]]) { SWAP_PTR(v1, v4); SWAP_PTR(v3, v5); }
    return Td[PA[*v3]] > Td[PA[*v4]] ? v4 : v3;
}

/* SsSort.cs:1275-1306 ss_pivot */
static inline IDX *FN(ss_pivot)(const uint8_t *Td, const IDX *PA, IDX *first, IDX *last)
{
    IDX t = (IDX)(last - first);
    IDX *middle = first + t / 2;
    if (t <= 512) {
        if (t <= 32) return FN(ss_median3)(Td, PA, first, middle, last - 1);
        t >>= 2;
        return FN(ss_median5)(Td, PA, first, first + t, middle, last - 1 - t, last - 1);
    }
    t >>= 3;
    first = FN(ss_median3)(Td, PA, first, first + t, first + (t << 1));
    middle = FN(ss_median3)(Td, PA, middle - t, middle, middle + t);
    last = FN(ss_median3)(Td, PA, last - 1 - (t << 1), last - 1 - t, last - 1);
    return FN(ss_median3)(Td, PA, first, middle, last);
}

/* SsSort.cs:1374-1432 ss_partition: substrings that end at `depth` go first, complemented */
static inline IDX *FN(ss_partition)(const IDX *PA, IDX *first, IDX *last, IDX depth)
{
    IDX *a, *b;
    IDX t;
    for (a = first - 1, b = last;;) {
        for (; (++a < b) && ((PA[*a] + depth) >= (PA[*a + 1] + 1));) *a = ~*a;
        for (; (a < --b) && ((PA[*b] + depth) < (PA[*b + 1] + 1));) {}
        if (b <= a) break;
        t = ~*b; *b = *a; *a = t;
    }
    if (first < a) *first = ~*first;
    return a;
}

typedef struct { IDX *a, *b, *c; IDX d; } FN(ss_item);

/* SsSort.cs:934-1269 ss_mintrosort: multikey introsort on T[PA[SA[i]] + depth] */
static void FN(ss_mintrosort)(const uint8_t *T, const IDX *PA, IDX *first, IDX *last, IDX depth)
{
    struct { IDX *a, *b; IDX c; int d; } stack[SS_STACK_SIZE];
    int ssize = 0;
#define SS_PUSH(A, B, C, D) do { assert(ssize < SS_STACK_SIZE); stack[ssize].a = (A); stack[ssize].b = (B); \
                                 stack[ssize].c = (C); stack[ssize++].d = (D); } while (0)
#define SS_POP(A, B, C, D) do { if (ssize == 0) return; (A) = stack[--ssize].a; (B) = stack[ssize].b; \
                                (C) = stack[ssize].c; (D) = stack[ssize].d; } while (0)
    const uint8_t *Td;
    IDX *a, *b, *c, *d, *e, *f;
    IDX s, t;
    int limit, v, x = 0;

    for (limit = FN(ilg)((IDX)(last - first));;) {
        if ((last - first) <= SS_INSERTIONSORT_THRESHOLD) {
            if (1 < (last - first)) FN(ss_insertionsort)(T, PA, first, last, depth);
            SS_POP(first, last, depth, limit);
            continue;
        }
        Td = T + depth;
        if (limit-- == 0) FN(ss_heapsort)(Td, PA, first, (IDX)(last - first));
        if (limit < 0) {
            for (a = first + 1, v = Td[PA[*first]]; a < last; ++a) {
                if ((x = Td[PA[*a]]) != v) {
                    if (1 < (a - first)) break;
                    v = x;
                    first = a;
                }
            }
            if (Td[PA[*first] - 1] < v) first = FN(ss_partition)(PA, first, a, depth);
            if ((a - first) <= (last - a)) {
                if (1 < (a - first)) {
                    SS_PUSH(a, last, depth, -1);
                    last = a; depth += 1; limit = FN(ilg)((IDX)(a - first));
                } else {
                    first = a; limit = -1;
                }
            } else {
                if (1 < (last - a)) {
                    SS_PUSH(first, a, depth + 1, FN(ilg)((IDX)(a - first)));
                    first = a; limit = -1;
                } else {
                    last = a; depth += 1; limit = FN(ilg)((IDX)(a - first));
                }
            }
            continue;
        }

        /* choose pivot */
        a = FN(ss_pivot)(Td, PA, first, last);
        v = Td[PA[*a]];
        SWAP_IDX(*first, *a);

        /* ternary partition */
        for (b = first; (++b < last) && ((x = Td[PA[*b]]) == v);) {}
        if (((a = b) < last) && (x < v)) {
            for (; (++b < last) && ((x = Td[PA[*b]]) <= v);)
                if (x == v) { SWAP_IDX(*b, *a); ++a; }
        }
        for (c = last; (b < --c) && ((x = Td[PA[*c]]) == v);) {}
        if ((b < (d = c)) && (x > v)) {
            for (; (b < --c) && ((x = Td[PA[*c]]) >= v);)
                if (x == v) { SWAP_IDX(*c, *d); --d; }
        }
        for (; b < c;) {
            SWAP_IDX(*b, *c);
            for (; (++b < c) && ((x = Td[PA[*b]]) <= v);)
                if (x == v) { SWAP_IDX(*b, *a); ++a; }
            for (; (b < --c) && ((x = Td[PA[*c]]) >= v);)
                if (x == v) { SWAP_IDX(*c, *d); --d; }
        }

        if (a <= d) {
            c = b - 1;
            if ((s = (IDX)(a - first)) > (t = (IDX)(b - a))) s = t;
            for (e = first, f = b - s; 0 < s; --s, ++e, ++f) SWAP_IDX(*e, *f);
            if ((s = (IDX)(d - c)) > (t = (IDX)(last - d - 1))) s = t;
            for (e = b, f = last - s; 0 < s; --s, ++e, ++f) SWAP_IDX(*e, *f);

            a = first + (b - a); c = last - (d - c);
            b = (v <= Td[PA[*a] - 1]) ? a : FN(ss_partition)(PA, a, c, depth);

            if ((a - first) <= (last - c)) {
                if ((last - c) <= (c - b)) {
                    SS_PUSH(b, c, depth + 1, FN(ilg)((IDX)(c - b)));
                    SS_PUSH(c, last, depth, limit);
                    last = a;
                } else if ((a - first) <= (c - b)) {
                    SS_PUSH(c, last, depth, limit);
                    SS_PUSH(b, c, depth + 1, FN(ilg)((IDX)(c - b)));
                    last = a;
                } else {
                    SS_PUSH(c, last, depth, limit);
                    SS_PUSH(first, a, depth, limit);
                    first = b; last = c; depth += 1; limit = FN(ilg)((IDX)(c - b));
                }
            } else {
                if ((a - first) <= (c - b)) {
                    SS_PUSH(b, c, depth + 1, FN(ilg)((IDX)(c - b)));
                    SS_PUSH(first, a, depth, limit);
                    first = c;
                } else if ((last - c) <= (c - b)) {
                    SS_PUSH(first, a, depth, limit);
                    SS_PUSH(b, c, depth + 1, FN(ilg)((IDX)(c - b)));
                    first = c;
                } else {
                    SS_PUSH(first, a, depth, limit);
                    SS_PUSH(c, last, depth, limit);
                    first = b; last = c; depth += 1; limit = FN(ilg)((IDX)(c - b));
                }
            }
        } else {
            limit += 1;
            if (Td[PA[*first] - 1] < v) {
                first = FN(ss_partition)(PA, first, last, depth);
                limit = FN(ilg)((IDX)(last - first));
            }
            depth += 1;
        }
    }
#undef SS_PUSH
#undef SS_POP
}

/* SsSort.cs:367-373 ss_blockswap */
static inline void FN(ss_blockswap)(IDX *a, IDX *b, IDX n)
{
    for (; 0 < n; --n, ++a, ++b) SWAP_IDX(*a, *b);
}

/* SsSort.cs:282-364 ss_rotate */
static inline void FN(ss_rotate)(IDX *first, IDX *middle, IDX *last)
{
    IDX *a, *b, t;
    IDX l, r;
    l = (IDX)(middle - first); r = (IDX)(last - middle);
    for (; (0 < l) && (0 < r);) {
        if (l == r) { FN(ss_blockswap)(first, middle, l); break; }
        if (l < r) {
            a = last - 1; b = middle - 1;
            t = *a;
            do {
                *a-- = *b; *b-- = *a;
                if (b < first) {
                    *a = t;
                    last = a;
                    if ((r -= l + 1) <= l) break;
                    a -= 1; b = middle - 1;
                    t = *a;
                }
            } while (1);
        } else {
            a = first; b = middle;
            t = *a;
            do {
                *a++ = *b; *b++ = *a;
                if (last <= b) {
                    *a = t;
                    first = a + 1;
                    if ((l -= r + 1) <= r) break;
                    a += 1; b = middle;
                    t = *a;
                }
            } while (1);
        }
    }
}

/* SsSort.cs:196-279 ss_inplacemerge */
static void FN(ss_inplacemerge)(const uint8_t *T, const IDX *PA, IDX *first, IDX *middle, IDX *last,
                                IDX depth)
{
    const IDX *p;
    IDX *a, *b;
    IDX len, half;
    int q, r, x;
    for (;;) {
        if (*(last - 1) < 0) { x = 1; p = PA + ~*(last - 1); }
        else { x = 0; p = PA + *(last - 1); }
        for (a = first, len = (IDX)(middle - first), half = len >> 1, r = -1; 0 < len;
             len = half, half >>= 1) {
            b = a + half;
            q = FN(ss_compare)(T, PA + ((0 <= *b) ? *b : ~*b), p, depth);
            if (q < 0) { a = b + 1; half -= (len & 1) ^ 1; }
            else r = q;
        }
        if (a < middle) {
            if (r == 0) *a = ~*a;
            FN(ss_rotate)(a, middle, last);
            last -= middle - a;
            middle = a;
            if (first == middle) break;
        }
        --last;
        if (x != 0) { while (*--last < 0) {} }
        if (middle == last) break;
    }
}

/* SsSort.cs:773-897 ss_mergeforward */
static void FN(ss_mergeforward)(const uint8_t *T, const IDX *PA, IDX *first, IDX *middle, IDX *last,
                                IDX *buf, IDX depth)
{
    IDX *a, *b, *c, *bufend;
    IDX t;
    int r;
    bufend = buf + (middle - first) - 1;
    FN(ss_blockswap)(buf, first, (IDX)(middle - first));
    for (t = *(a = first), b = buf, c = middle;;) {
        r = FN(ss_compare)(T, PA + *b, PA + *c, depth);
        if (r < 0) {
            do {
                *a++ = *b;
                if (bufend <= b) { *bufend = t; return; }
                *b++ = *a;
            } while (*b < 0);
        } else if (r > 0) {
            do {
                *a++ = *c; *c++ = *a;
                if (last <= c) {
                    while (b < bufend) { *a++ = *b; *b++ = *a; }
                    *a = *b; *b = t;
                    return;
                }
            } while (*c < 0);
        } else {
            *c = ~*c;
            do {
                *a++ = *b;
                if (bufend <= b) { *bufend = t; return; }
                *b++ = *a;
            } while (*b < 0);
            do {
                *a++ = *c; *c++ = *a;
                if (last <= c) {
                    while (b < bufend) { *a++ = *b; *b++ = *a; }
                    *a = *b; *b = t;
                    return;
                }
            } while (*c < 0);
        }
    }
}

/* SsSort.cs:563-770 ss_mergebackward */
static void FN(ss_mergebackward)(const uint8_t *T, const IDX *PA, IDX *first, IDX *middle, IDX *last,
                                 IDX *buf, IDX depth)
{
    const IDX *p1, *p2;
    IDX *a, *b, *c, *bufend;
    IDX t;
    int r, x;
    bufend = buf + (last - middle) - 1;
    FN(ss_blockswap)(buf, middle, (IDX)(last - middle));

    x = 0;
    if (*bufend < 0) { p1 = PA + ~*bufend; x |= 1; } else { p1 = PA + *bufend; }
    if (*(middle - 1) < 0) { p2 = PA + ~*(middle - 1); x |= 2; } else { p2 = PA + *(middle - 1); }
    for (t = *(a = last - 1), b = bufend, c = middle - 1;;) {
        r = FN(ss_compare)(T, p1, p2, depth);
        if (0 < r) {
            if (x & 1) { do { *a-- = *b; *b-- = *a; } while (*b < 0); x ^= 1; }
            *a-- = *b;
            if (b <= buf) { *buf = t; break; }
            *b-- = *a;
            if (*b < 0) { p1 = PA + ~*b; x |= 1; } else { p1 = PA + *b; }
        } else if (r < 0) {
            if (x & 2) { do { *a-- = *c; *c-- = *a; } while (*c < 0); x ^= 2; }
            *a-- = *c; *c-- = *a;
            if (c < first) {
                while (buf < b) { *a-- = *b; *b-- = *a; }
                *a = *b; *b = t;
                break;
            }
            if (*c < 0) { p2 = PA + ~*c; x |= 2; } else { p2 = PA + *c; }
        } else {
            if (x & 1) { do { *a-- = *b; *b-- = *a; } while (*b < 0); x ^= 1; }
            *a-- = ~*b;
            if (b <= buf) { *buf = t; break; }
            *b-- = *a;
            if (x & 2) { do { *a-- = *c; *c-- = *a; } while (*c < 0); x ^= 2; }
            *a-- = *c; *c-- = *a;
            if (c < first) {
                while (buf < b) { *a-- = *b; *b-- = *a; }
                *a = *b; *b = t;
                break;
            }
            if (*b < 0) { p1 = PA + ~*b; x |= 1; } else { p1 = PA + *b; }
            if (*c < 0) { p2 = PA + ~*c; x |= 2; } else { p2 = PA + *c; }
        }
    }
}

/* SsSort.cs:376-560 ss_swapmerge: divide-and-conquer merge with a bounded buffer */
static void FN(ss_swapmerge)(const uint8_t *T, const IDX *PA, IDX *first, IDX *middle, IDX *last,
                             IDX *buf, IDX bufsize, IDX depth)
{
#define GETIDX(a) ((0 <= (a)) ? (a) : (~(a)))
#define MERGE_CHECK(a, b, c)                                                                     \
    do {                                                                                         \
        if (((c) & 1) || (((c) & 2) && (FN(ss_compare)(T, PA + GETIDX(*((a) - 1)), PA + *(a), depth) == 0))) \
            *(a) = ~*(a);                                                                        \
        if (((c) & 4) && ((FN(ss_compare)(T, PA + GETIDX(*((b) - 1)), PA + *(b), depth) == 0)))  \
            *(b) = ~*(b);                                                                        \
    } while (0)
    struct { IDX *a, *b, *c; int d; } stack[SS_MERGE_STACK_SIZE];
    int ssize = 0;
#define SM_PUSH(A, B, C, D) do { assert(ssize < SS_MERGE_STACK_SIZE); stack[ssize].a = (A); stack[ssize].b = (B); \
                                 stack[ssize].c = (C); stack[ssize++].d = (D); } while (0)
#define SM_POP(A, B, C, D) do { if (ssize == 0) return; (A) = stack[--ssize].a; (B) = stack[ssize].b; \
                                (C) = stack[ssize].c; (D) = stack[ssize].d; } while (0)
    IDX *l, *r, *lm, *rm;
    IDX m, len, half;
    int check, next;

    for (check = 0;;) {
        if ((last - middle) <= bufsize) {
            if ((first < middle) && (middle < last))
                FN(ss_mergebackward)(T, PA, first, middle, last, buf, depth);
            MERGE_CHECK(first, last, check);
            SM_POP(first, middle, last, check);
            continue;
        }
        if ((middle - first) <= bufsize) {
            if (first < middle) FN(ss_mergeforward)(T, PA, first, middle, last, buf, depth);
            MERGE_CHECK(first, last, check);
            SM_POP(first, middle, last, check);
            continue;
        }
        for (m = 0, len = (IDX)((middle - first) < (last - middle) ? (middle - first) : (last - middle)),
            half = len >> 1;
             0 < len; len = half, half >>= 1) {
            if (FN(ss_compare)(T, PA + GETIDX(*(middle + m + half)),
                               PA + GETIDX(*(middle - m - half - 1)), depth) < 0) {
                m += half + 1;
                half -= (len & 1) ^ 1;
            }
        }
        if (0 < m) {
            lm = middle - m; rm = middle + m;
            FN(ss_blockswap)(lm, middle, m);
            l = r = middle; next = 0;
            if (rm < last) {
                if (*rm < 0) {
                    *rm = ~*rm;
                    if (first < lm) { for (; *--l < 0;) {} next |= 4; }
                    next |= 1;
                } else if (first < lm) {
                    for (; *r < 0; ++r) {}
                    next |= 2;
                }
            }
            if ((l - first) <= (last - r)) {
                SM_PUSH(r, rm, last, (next & 3) | (check & 4));
                middle = lm; last = l; check = (check & 3) | (next & 4);
            } else {
                if ((next & 2) && (r == middle)) next ^= 6;
                SM_PUSH(first, lm, l, (check & 3) | (next & 4));
                first = r; middle = rm; check = (next & 3) | (check & 4);
            }
        } else {
            if (FN(ss_compare)(T, PA + GETIDX(*(middle - 1)), PA + *middle, depth) == 0)
                *middle = ~*middle;
            MERGE_CHECK(first, last, check);
            SM_POP(first, middle, last, check);
        }
    }
#undef SM_PUSH
#undef SM_POP
#undef MERGE_CHECK
#undef GETIDX
}

/* SsSort.cs:23-149 sssort: sort one (c0,c1) bucket of B* substrings */
static void FN(sssort)(const uint8_t *T, const IDX *PA, IDX *first, IDX *last, IDX *buf, IDX bufsize,
                       IDX depth, IDX n, int lastsuffix)
{
    IDX *a, *b, *middle, *curbuf;
    IDX j, k, curbufsize, limit;
    IDX i;

    if (lastsuffix != 0) ++first;

    if ((bufsize < SS_BLOCKSIZE) && (bufsize < (last - first)) &&
        (bufsize < (limit = FN(ss_isqrt)((IDX)(last - first))))) {
        if (SS_BLOCKSIZE < limit) limit = SS_BLOCKSIZE;
        buf = middle = last - limit; bufsize = limit;
    } else {
        middle = last; limit = 0;
    }
    for (a = first, i = 0; SS_BLOCKSIZE < (middle - a); a += SS_BLOCKSIZE, ++i) {
        FN(ss_mintrosort)(T, PA, a, a + SS_BLOCKSIZE, depth);
        curbufsize = (IDX)(last - (a + SS_BLOCKSIZE));
        curbuf = a + SS_BLOCKSIZE;
        if (curbufsize <= bufsize) { curbufsize = bufsize; curbuf = buf; }
        for (b = a, k = SS_BLOCKSIZE, j = i; j & 1; b -= k, k <<= 1, j >>= 1)
            FN(ss_swapmerge)(T, PA, b - k, b, b + k, curbuf, curbufsize, depth);
    }
    FN(ss_mintrosort)(T, PA, a, middle, depth);
    for (k = SS_BLOCKSIZE; i != 0; k <<= 1, i >>= 1) {
        if (i & 1) {
            FN(ss_swapmerge)(T, PA, a - k, a, middle, buf, bufsize, depth);
            a -= k;
        }
    }
    if (limit != 0) {
        FN(ss_mintrosort)(T, PA, middle, last, depth);
        FN(ss_inplacemerge)(T, PA, first, middle, last, depth);
    }

    if (lastsuffix != 0) {
        /* SsSort.cs:129-148: insert the last type B* suffix */
        IDX PAi[2];
        PAi[0] = PA[*(first - 1)]; PAi[1] = n - 2;
        for (a = first, i = *(first - 1);
             (a < last) && ((*a < 0) || (0 < FN(ss_compare)(T, &(PAi[0]), PA + *a, depth))); ++a)
            *(a - 1) = *a;
        *(a - 1) = i;
    }
}

/* ------------------------------------------------------------------ trsort */

/* TrSort.cs:954-1006 tr_insertionsort */
static void FN(tr_insertionsort)(const IDX *ISAd, IDX *first, IDX *last)
{
    IDX *a, *b;
    IDX t, r;
    for (a = first + 1; a < last; ++a) {
        for (t = *a, b = a - 1; 0 > (r = ISAd[t] - ISAd[*b]);) {
            do { *(b + 1) = *b; } while ((first <= --b) && (*b < 0));
            if (b < first) break;
        }
        if (r == 0) *b = ~*b;
        *(b + 1) = t;
    }
}

/* TrSort.cs:908-949 tr_fixdown */
static inline void FN(tr_fixdown)(const IDX *ISAd, IDX *SA, IDX i, IDX size)
{
    IDX j, k, v, c, d, e;
    for (v = SA[i], c = ISAd[v]; (j = 2 * i + 1) < size; SA[i] = SA[k], i = k) {
        d = ISAd[SA[k = j++]];
        if (d < (e = ISAd[SA[j]])) { k = j; d = e; }
        if (d <= c) break;
    }
    SA[i] = v;
}

/* TrSort.cs:865-905 tr_heapsort */
static void FN(tr_heapsort)(const IDX *ISAd, IDX *SA, IDX size)
{
    IDX i, m, t;
    m = size;
    if ((size % 2) == 0) {
        m--;
        if (ISAd[SA[m / 2]] < ISAd[SA[m]]) SWAP_IDX(SA[m], SA[m / 2]);
    }
    for (i = m / 2 - 1; 0 <= i; --i) FN(tr_fixdown)(ISAd, SA, i, m);
    if ((size % 2) == 0) { SWAP_IDX(SA[0], SA[m]); FN(tr_fixdown)(ISAd, SA, 0, m); }
    for (i = m - 1; 0 < i; --i) {
        t = SA[0]; SA[0] = SA[i];
        FN(tr_fixdown)(ISAd, SA, 0, i);
        SA[i] = t;
    }
}

/* TrSort.cs:839-862 tr_median3 */
static inline IDX *FN(tr_median3)(const IDX *ISAd, IDX *v1, IDX *v2, IDX *v3)
{
    if (ISAd[*v1] > ISAd[*v2]) SWAP_PTR(v1, v2);
    if (ISAd[*v2] > ISAd[*v3]) return ISAd[*v1] > ISAd[*v3] ? v1 : v3;
    return v2;
}

/* TrSort.cs:799-833 tr_median5 */
static inline IDX *FN(tr_median5)(const IDX *ISAd, IDX *v1, IDX *v2, IDX *v3, IDX *v4, IDX *v5)
{
    if (ISAd[*v2] > ISAd[*v3]) SWAP_PTR(v2, v3);
    if (ISAd[*v4] > ISAd[*v5]) SWAP_PTR(v4, v5);
    if (ISAd[*v2] > ISAd[*v4]) { SWAP_PTR(v2, v4); SWAP_PTR(v3, v5); }
    if (ISAd[*v1] > ISAd[*v3]) SWAP_PTR(v1, v3);
    if (ISAd[*v1] > ISAd[*v4]) { SWAP_PTR(v1, v4); SWAP_PTR(v3, v5); }
    return ISAd[*v3] > ISAd[*v4] ? v4 : v3;
}

/* TrSort.cs:771-793 tr_pivot */
static inline IDX *FN(tr_pivot)(const IDX *ISAd, IDX *first, IDX *last)
{
    IDX t = (IDX)(last - first);
    IDX *middle = first + t / 2;
    if (t <= 512) {
        if (t <= 32) return FN(tr_median3)(ISAd, first, middle, last - 1);
        t >>= 2;
        return FN(tr_median5)(ISAd, first, first + t, middle, last - 1 - t, last - 1);
    }
    t >>= 3;
    first = FN(tr_median3)(ISAd, first, first + t, first + (t << 1));
    middle = FN(tr_median3)(ISAd, middle - t, middle, middle + t);
    last = FN(tr_median3)(ISAd, last - 1 - (t << 1), last - 1 - t, last - 1);
    return FN(tr_median3)(ISAd, first, middle, last);
}

/* Budget.cs:2-34 */
typedef struct { IDX chance, remain, incval, count; } FN(budget_t);

static inline void FN(budget_init)(FN(budget_t) *b, IDX chance, IDX incval)
{
    b->chance = chance; b->remain = b->incval = incval; b->count = 0;
}

static inline int FN(budget_check)(FN(budget_t) *b, IDX size)
{
    if (size <= b->remain) { b->remain -= size; return 1; }
    if (b->chance == 0) { b->count += size; return 0; }
    b->remain += b->incval - size;
    b->chance -= 1;
    return 1;
}

/* TrSort.cs:1148-1324 tr_partition: ternary split of [first,last) on ISAd[.] vs v */
static inline void FN(tr_partition)(const IDX *ISAd, IDX *first, IDX *middle, IDX *last, IDX **pa,
                                    IDX **pb, IDX v)
{
    IDX *a, *b, *c, *d, *e, *f;
    IDX t, s, x = 0;
    for (b = middle - 1; (++b < last) && ((x = ISAd[*b]) == v);) {}
    if (((a = b) < last) && (x < v)) {
        for (; (++b < last) && ((x = ISAd[*b]) <= v);)
            if (x == v) { SWAP_IDX(*b, *a); ++a; }
    }
    for (c = last; (b < --c) && ((x = ISAd[*c]) == v);) {}
    if ((b < (d = c)) && (x > v)) {
        for (; (b < --c) && ((x = ISAd[*c]) >= v);)
            if (x == v) { SWAP_IDX(*c, *d); --d; }
    }
    for (; b < c;) {
        SWAP_IDX(*b, *c);
        for (; (++b < c) && ((x = ISAd[*b]) <= v);)
            if (x == v) { SWAP_IDX(*b, *a); ++a; }
        for (; (b < --c) && ((x = ISAd[*c]) >= v);)
            if (x == v) { SWAP_IDX(*c, *d); --d; }
    }
    if (a <= d) {
        c = b - 1;
        if ((s = (IDX)(a - first)) > (t = (IDX)(b - a))) s = t;
        for (e = first, f = b - s; 0 < s; --s, ++e, ++f) SWAP_IDX(*e, *f);
        if ((s = (IDX)(d - c)) > (t = (IDX)(last - d - 1))) s = t;
        for (e = b, f = last - s; 0 < s; --s, ++e, ++f) SWAP_IDX(*e, *f);
        first += (b - a); last -= (d - c);
    }
    *pa = first; *pb = last;
}

/* TrSort.cs:1092-1142 tr_copy */
static void FN(tr_copy)(IDX *ISA, const IDX *SA, IDX *first, IDX *a, IDX *b, IDX *last, IDX depth)
{
    IDX *c, *d, *e;
    IDX s, v;
    v = (IDX)(b - SA - 1);
    for (c = first, d = a - 1; c <= d; ++c) {
        if ((0 <= (s = *c - depth)) && (ISA[s] == v)) {
            *++d = s;
            ISA[s] = (IDX)(d - SA);
        }
    }
    for (c = last - 1, e = d + 1, d = b; e < d; --c) {
        if ((0 <= (s = *c - depth)) && (ISA[s] == v)) {
            *--d = s;
            ISA[s] = (IDX)(d - SA);
        }
    }
}

/* TrSort.cs:1008-1087 tr_partialcopy */
static void FN(tr_partialcopy)(IDX *ISA, const IDX *SA, IDX *first, IDX *a, IDX *b, IDX *last,
                               IDX depth)
{
    IDX *c, *d, *e;
    IDX s, v, rank, lastrank, newrank = -1;
    v = (IDX)(b - SA - 1);
    lastrank = -1;
    for (c = first, d = a - 1; c <= d; ++c) {
        if ((0 <= (s = *c - depth)) && (ISA[s] == v)) {
            *++d = s;
            rank = ISA[s + depth];
            if (lastrank != rank) { lastrank = rank; newrank = (IDX)(d - SA); }
            ISA[s] = newrank;
        }
    }
    lastrank = -1;
    for (e = d; first <= e; --e) {
        rank = ISA[*e];
        if (lastrank != rank) { lastrank = rank; newrank = (IDX)(e - SA); }
        if (newrank != rank) ISA[*e] = newrank;
    }
    lastrank = -1;
    for (c = last - 1, e = d + 1, d = b; e < d; --c) {
        if ((0 <= (s = *c - depth)) && (ISA[s] == v)) {
            *--d = s;
            rank = ISA[s + depth];
            if (lastrank != rank) { lastrank = rank; newrank = (IDX)(d - SA); }
            ISA[s] = newrank;
        }
    }
}

/* TrSort.cs:148-765 tr_introsort */
static void FN(tr_introsort)(IDX *ISA, const IDX *ISAd, IDX *SA, IDX *first, IDX *last,
                             FN(budget_t) *budget)
{
    struct { const IDX *a; IDX *b, *c; int d, e; } stack[TR_STACK_SIZE];
    int ssize = 0;
#define TR_PUSH(A, B, C, D, E) do { assert(ssize < TR_STACK_SIZE); stack[ssize].a = (A); stack[ssize].b = (B); \
                                    stack[ssize].c = (C); stack[ssize].d = (D); stack[ssize++].e = (E); } while (0)
#define TR_POP(A, B, C, D, E) do { if (ssize == 0) return; (A) = stack[--ssize].a; (B) = stack[ssize].b; \
                                   (C) = stack[ssize].c; (D) = stack[ssize].d; (E) = stack[ssize].e; } while (0)
    IDX *a = NULL, *b = NULL, *c;
    IDX v, x = 0;
    IDX incr = (IDX)(ISAd - ISA);
    int limit, next;
    int trlink = -1;

    for (limit = FN(ilg)((IDX)(last - first));;) {
        if (limit < 0) {
            if (limit == -1) {
                /* tandem repeat partition (TrSort.cs:172-282) */
                FN(tr_partition)(ISAd - incr, first, first, last, &a, &b, (IDX)(last - SA - 1));
                /* update ranks */
                if (a < last) { for (c = first, v = (IDX)(a - SA - 1); c < a; ++c) ISA[*c] = v; }
                if (b < last) { for (c = a, v = (IDX)(b - SA - 1); c < b; ++c) ISA[*c] = v; }
                /* push */
                if (1 < (b - a)) {
                    TR_PUSH(NULL, a, b, 0, 0);
                    TR_PUSH(ISAd - incr, first, last, -2, trlink);
                    trlink = ssize - 2;
                }
                if ((a - first) <= (last - b)) {
                    if (1 < (a - first)) {
                        TR_PUSH(ISAd, b, last, FN(ilg)((IDX)(last - b)), trlink);
                        last = a; limit = FN(ilg)((IDX)(a - first));
                    } else if (1 < (last - b)) {
                        first = b; limit = FN(ilg)((IDX)(last - b));
                    } else {
                        TR_POP(ISAd, first, last, limit, trlink);
                    }
                } else {
                    if (1 < (last - b)) {
                        TR_PUSH(ISAd, first, a, FN(ilg)((IDX)(a - first)), trlink);
                        first = b; limit = FN(ilg)((IDX)(last - b));
                    } else if (1 < (a - first)) {
                        last = a; limit = FN(ilg)((IDX)(a - first));
                    } else {
                        TR_POP(ISAd, first, last, limit, trlink);
                    }
                }
            } else if (limit == -2) {
                /* tandem repeat copy (TrSort.cs:283-309) */
                a = stack[--ssize].b; b = stack[ssize].c;
                if (stack[ssize].d == 0) {
                    FN(tr_copy)(ISA, SA, first, a, b, last, (IDX)(ISAd - ISA));
                } else {
                    if (0 <= trlink) stack[trlink].d = -1;
                    FN(tr_partialcopy)(ISA, SA, first, a, b, last, (IDX)(ISAd - ISA));
                }
                TR_POP(ISAd, first, last, limit, trlink);
            } else {
                /* sorted partition (TrSort.cs:310-438) */
                if (0 <= *first) {
                    a = first;
                    do { ISA[*a] = (IDX)(a - SA); } while ((++a < last) && (0 <= *a));
                    first = a;
                }
                if (first < last) {
                    a = first;
                    do { *a = ~*a; } while (*++a < 0);
                    next = (ISA[*a] != ISAd[*a]) ? FN(ilg)((IDX)(a - first + 1)) : -1;
                    if (++a < last) { for (b = first, v = (IDX)(a - SA - 1); b < a; ++b) ISA[*b] = v; }

                    /* push */
                    if (FN(budget_check)(budget, (IDX)(a - first))) {
                        if ((a - first) <= (last - a)) {
                            TR_PUSH(ISAd, a, last, -3, trlink);
                            ISAd += incr; last = a; limit = next;
                        } else {
                            if (1 < (last - a)) {
                                TR_PUSH(ISAd + incr, first, a, next, trlink);
                                first = a; limit = -3;
                            } else {
                                ISAd += incr; last = a; limit = next;
                            }
                        }
                    } else {
                        if (0 <= trlink) stack[trlink].d = -1;
                        if (1 < (last - a)) {
                            first = a; limit = -3;
                        } else {
                            TR_POP(ISAd, first, last, limit, trlink);
                        }
                    }
                } else {
                    TR_POP(ISAd, first, last, limit, trlink);
                }
            }
            continue;
        }

        if ((last - first) <= TR_INSERTIONSORT_THRESHOLD) {
            FN(tr_insertionsort)(ISAd, first, last);
            limit = -3;
            continue;
        }

        if (limit-- == 0) {
            FN(tr_heapsort)(ISAd, first, (IDX)(last - first));
            for (a = last - 1; first < a; a = b) {
                for (x = ISAd[*a], b = a - 1; (first <= b) && (ISAd[*b] == x); --b) *b = ~*b;
            }
            limit = -3;
            continue;
        }

        /* choose pivot */
        a = FN(tr_pivot)(ISAd, first, last);
        SWAP_IDX(*first, *a);
        v = ISAd[*first];

        /* partition */
        FN(tr_partition)(ISAd, first, first + 1, last, &a, &b, v);
        if ((last - first) != (b - a)) {
            next = (ISA[*a] != v) ? FN(ilg)((IDX)(b - a)) : -1;

            /* update ranks */
            for (c = first, v = (IDX)(a - SA - 1); c < a; ++c) ISA[*c] = v;
            if (b < last) { for (c = a, v = (IDX)(b - SA - 1); c < b; ++c) ISA[*c] = v; }

            /* push */
            if ((1 < (b - a)) && (FN(budget_check)(budget, (IDX)(b - a)))) {
                if ((a - first) <= (last - b)) {
                    if ((last - b) <= (b - a)) {
                        if (1 < (a - first)) {
                            TR_PUSH(ISAd + incr, a, b, next, trlink);
                            TR_PUSH(ISAd, b, last, limit, trlink);
                            last = a;
                        } else if (1 < (last - b)) {
                            TR_PUSH(ISAd + incr, a, b, next, trlink);
                            first = b;
                        } else {
                            ISAd += incr; first = a; last = b; limit = next;
                        }
                    } else if ((a - first) <= (b - a)) {
                        if (1 < (a - first)) {
                            TR_PUSH(ISAd, b, last, limit, trlink);
                            TR_PUSH(ISAd + incr, a, b, next, trlink);
                            last = a;
                        } else {
                            TR_PUSH(ISAd, b, last, limit, trlink);
                            ISAd += incr; first = a; last = b; limit = next;
                        }
                    } else {
                        TR_PUSH(ISAd, b, last, limit, trlink);
                        TR_PUSH(ISAd, first, a, limit, trlink);
                        ISAd += incr; first = a; last = b; limit = next;
                    }
                } else {
                    if ((a - first) <= (b - a)) {
                        if (1 < (last - b)) {
                            TR_PUSH(ISAd + incr, a, b, next, trlink);
                            TR_PUSH(ISAd, first, a, limit, trlink);
                            first = b;
                        } else if (1 < (a - first)) {
                            TR_PUSH(ISAd + incr, a, b, next, trlink);
                            last = a;
                        } else {
                            ISAd += incr; first = a; last = b; limit = next;
                        }
                    } else if ((last - b) <= (b - a)) {
                        if (1 < (last - b)) {
                            TR_PUSH(ISAd, first, a, limit, trlink);
                            TR_PUSH(ISAd + incr, a, b, next, trlink);
                            first = b;
                        } else {
                            TR_PUSH(ISAd, first, a, limit, trlink);
                            ISAd += incr; first = a; last = b; limit = next;
                        }
                    } else {
                        TR_PUSH(ISAd, first, a, limit, trlink);
                        TR_PUSH(ISAd, b, last, limit, trlink);
                        ISAd += incr; first = a; last = b; limit = next;
                    }
                }
            } else {
                if ((1 < (b - a)) && (0 <= trlink)) stack[trlink].d = -1;
                if ((a - first) <= (last - b)) {
                    if (1 < (a - first)) {
                        TR_PUSH(ISAd, b, last, limit, trlink);
                        last = a;
                    } else if (1 < (last - b)) {
                        first = b;
                    } else {
                        TR_POP(ISAd, first, last, limit, trlink);
                    }
                } else {
                    if (1 < (last - b)) {
                        TR_PUSH(ISAd, first, a, limit, trlink);
                        first = b;
                    } else if (1 < (a - first)) {
                        last = a;
                    } else {
                        TR_POP(ISAd, first, last, limit, trlink);
                    }
                }
            }
        } else {
            if (FN(budget_check)(budget, (IDX)(last - first))) {
                limit = FN(ilg)((IDX)(last - first));
                ISAd += incr;
            } else {
                if (0 <= trlink) stack[trlink].d = -1;
                TR_POP(ISAd, first, last, limit, trlink);
            }
        }
    }
#undef TR_PUSH
#undef TR_POP
}

/* TrSort.cs:19-102 trsort: rank doubling over the B* suffixes until every group is sorted */
static void FN(trsort)(IDX *ISA, IDX *SA, IDX n, IDX depth)
{
    IDX *ISAd;
    IDX *first, *last;
    FN(budget_t) budget;
    IDX t, skip, unsorted;

    FN(budget_init)(&budget, (IDX)(FN(ilg)(n) * 2 / 3), n);
    for (ISAd = ISA + depth; -n < *SA; ISAd += ISAd - ISA) {
        first = SA;
        skip = 0;
        unsorted = 0;
        do {
            if ((t = *first) < 0) {
                first -= t; skip += t;
            } else {
                if (skip != 0) { *(first + skip) = skip; skip = 0; }
                last = SA + ISA[t] + 1;
                if (1 < (last - first)) {
                    budget.count = 0;
                    FN(tr_introsort)(ISA, ISAd, SA, first, last, &budget);
                    if (budget.count != 0) unsorted += budget.count;
                    else skip = (IDX)(first - last);
                } else if ((last - first) == 1) {
                    skip = -1;
                }
                first = last;
            }
        } while (first < (SA + n));
        if (skip != 0) *(first + skip) = skip;
        if (unsorted == 0) break;
    }
}

/* ------------------------------------------------------------------ driver */

#define BUCKET_A(c0) bucket_A[(c0)]
#define BUCKET_B(c0, c1) (bucket_B[((c1) << 8) | (c0)])          /* DivSufSort.cs:175-184 BBucket */
#define BUCKET_BSTAR(c0, c1) (bucket_B[((c0) << 8) | (c1)])      /* DivSufSort.cs:163-172 BStarBucket */

/* DivSufSort.cs:186-511 sort_typeBstar */
static IDX FN(sort_typeBstar)(const uint8_t *T, IDX *SA, IDX *bucket_A, IDX *bucket_B, IDX n,
                              double *phase)
{
    IDX *PAb, *ISAb, *buf;
    IDX i, j, k, t, m, bufsize;
    int c0, c1;
    double t0 = dss_now();

    /* :202-267 count A / B / B* suffixes, store B* positions at the tail of SA */
    for (i = n - 1, m = n, c0 = T[n - 1]; 0 <= i;) {
        do { ++BUCKET_A(c1 = c0); } while ((0 <= --i) && ((c0 = T[i]) >= c1));
        if (0 <= i) {
            ++BUCKET_BSTAR(c0, c1);
            SA[--m] = i;
            for (--i, c1 = c0; (0 <= i) && ((c0 = T[i]) <= c1); --i, c1 = c0) ++BUCKET_B(c0, c1);
        }
    }
    m = n - m;

    /* :272-290 bucket start / end points */
    for (c0 = 0, i = 0, j = 0; c0 < ALPHABET_SIZE; ++c0) {
        t = i + BUCKET_A(c0);
        BUCKET_A(c0) = i + j;
        i = t + BUCKET_B(c0, c0);
        for (c1 = c0 + 1; c1 < ALPHABET_SIZE; ++c1) {
            j += BUCKET_BSTAR(c0, c1);
            BUCKET_BSTAR(c0, c1) = j;
            i += BUCKET_B(c0, c1);
        }
    }
    phase[0] += dss_now() - t0;

    if (0 < m) {
        /* :294-310 sort B* suffixes by their first two characters */
        t0 = dss_now();
        PAb = SA + n - m; ISAb = SA + m;
        for (i = m - 2; 0 <= i; --i) {
            t = PAb[i]; c0 = T[t]; c1 = T[t + 1];
            SA[--BUCKET_BSTAR(c0, c1)] = i;
        }
        t = PAb[m - 1]; c0 = T[t]; c1 = T[t + 1];
        SA[--BUCKET_BSTAR(c0, c1)] = m - 1;
        phase[0] += dss_now() - t0;

        /* :312-342 sort B* substrings bucket by bucket */
        t0 = dss_now();
        buf = SA + m; bufsize = n - (2 * m);
        for (c0 = ALPHABET_SIZE - 2, j = m; 0 < j; --c0) {
            for (c1 = ALPHABET_SIZE - 1; c0 < c1; j = i, --c1) {
                i = BUCKET_BSTAR(c0, c1);
                if (1 < (j - i))
                    FN(sssort)(T, PAb, SA + i, SA + j, buf, bufsize, 2, n, *(SA + i) == (m - 1));
            }
        }
        phase[1] += dss_now() - t0;

        /* :344-386 ranks of B* substrings */
        t0 = dss_now();
        for (i = m - 1; 0 <= i; --i) {
            if (0 <= SA[i]) {
                j = i;
                do { ISAb[SA[i]] = i; } while ((0 <= --i) && (0 <= SA[i]));
                SA[i + 1] = i - j;
                if (i <= 0) break;
            }
            j = i;
            do { ISAb[SA[i] = ~SA[i]] = j; } while (SA[--i] < 0);
            ISAb[SA[i]] = j;
        }

        /* :388-392 inverse suffix array of the B* suffixes */
        FN(trsort)(ISAb, SA, m, 1);
        phase[2] += dss_now() - t0;

        /* :394-461 sorted order of B* suffixes */
        t0 = dss_now();
        for (i = n - 1, j = m, c0 = T[n - 1]; 0 <= i;) {
            for (--i, c1 = c0; (0 <= i) && ((c0 = T[i]) >= c1); --i, c1 = c0) {}
            if (0 <= i) {
                t = i;
                for (--i, c1 = c0; (0 <= i) && ((c0 = T[i]) <= c1); --i, c1 = c0) {}
                SA[ISAb[--j]] = ((t == 0) || (1 < (t - i))) ? t : ~t;
            }
        }

        /* :463-507 bucket boundaries; move B* suffixes into place */
        BUCKET_B(ALPHABET_SIZE - 1, ALPHABET_SIZE - 1) = n;
        for (c0 = ALPHABET_SIZE - 2, k = m - 1; 0 <= c0; --c0) {
            i = BUCKET_A(c0 + 1) - 1;
            for (c1 = ALPHABET_SIZE - 1; c0 < c1; --c1) {
                t = i - BUCKET_B(c0, c1);
                BUCKET_B(c0, c1) = i;
                for (i = t, j = BUCKET_BSTAR(c0, c1); j <= k; --i, --k) SA[i] = SA[k];
            }
            BUCKET_BSTAR(c0, c0 + 1) = i - BUCKET_B(c0, c0) + 1;
            BUCKET_B(c0, c0) = i;
        }
        phase[3] += dss_now() - t0;
    }
    return m;
}

/* DivSufSort.cs:44-153 construct_SA: induce B from B*, then A from B */
static void FN(construct_SA)(const uint8_t *T, IDX *SA, IDX *bucket_A, IDX *bucket_B, IDX n, IDX m)
{
    IDX *i, *j, *k;
    IDX s;
    int c0, c1, c2;

    if (0 < m) {
        /* :55-107 right-to-left scan */
        for (c1 = ALPHABET_SIZE - 2; 0 <= c1; --c1) {
            for (i = SA + BUCKET_BSTAR(c1, c1 + 1), j = SA + BUCKET_A(c1 + 1) - 1, k = NULL, c2 = -1;
                 i <= j; --j) {
                if (0 < (s = *j)) {
                    *j = ~s;
                    c0 = T[--s];
                    if ((0 < s) && (T[s - 1] > c0)) s = ~s;
                    if (c0 != c2) {
                        if (0 <= c2) BUCKET_B(c2, c1) = (IDX)(k - SA);
                        k = SA + BUCKET_B(c2 = c0, c1);
                    }
                    *k-- = s;
                } else {
                    *j = ~s;
                }
            }
        }
    }

    /* :110-152 left-to-right scan */
    k = SA + BUCKET_A(c2 = T[n - 1]);
    *k++ = (T[n - 2] < c2) ? ~(n - 1) : (n - 1);
    for (i = SA, j = SA + n; i < j; ++i) {
        if (0 < (s = *i)) {
            c0 = T[--s];
            if ((s == 0) || (T[s - 1] < c0)) s = ~s;
            if (c0 != c2) {
                BUCKET_A(c2) = (IDX)(k - SA);
                k = SA + BUCKET_A(c2 = c0);
            }
            *k++ = s;
        } else {
            *i = ~s;
        }
    }
}

/* LibDivSufSort.cs:12-29 + DivSufSort.cs:18-42 divsufsort */
int32_t FN(dq_oracle_divsufsort)(const uint8_t *T, IDX *SA, int64_t n64)
{
    IDX *bucket_A, *bucket_B;
    IDX m, n = (IDX)n64;
    double *phase = dss_phase;

    for (int p = 0; p < 5; ++p) phase[p] = 0;
    if (n64 < 0 || (n64 > 0 && (T == NULL || SA == NULL))) return -1;
    if ((int64_t)n != n64 || (sizeof(IDX) == 4 && n64 > 0x7fffffffLL)) return -1;
    /* DivSufSort.cs:22-38 */
    if (n == 0) return 0;
    if (n == 1) { SA[0] = 0; return 0; }
    if (n == 2) {
        m = (T[0] < T[1]);
        SA[m ^ 1] = 0; SA[m] = 1;
        return 0;
    }
    /* DivSufSort.cs:190-192: both bucket arrays MUST be zeroed */
    bucket_A = (IDX *)calloc(BUCKET_A_SIZE, sizeof(IDX));
    bucket_B = (IDX *)calloc(BUCKET_B_SIZE, sizeof(IDX));
    if (bucket_A == NULL || bucket_B == NULL) { free(bucket_A); free(bucket_B); return -2; }
    m = FN(sort_typeBstar)(T, SA, bucket_A, bucket_B, n, phase);
    {
        double t0 = dss_now();
        FN(construct_SA)(T, SA, bucket_A, bucket_B, n, m);
        phase[4] += dss_now() - t0;
    }
    free(bucket_B);
    free(bucket_A);
    return 0;
}

#undef BUCKET_A
#undef BUCKET_B
#undef BUCKET_BSTAR
#undef SWAP_IDX
#undef SWAP_PTR
#undef FN
#undef DSS_CAT
#undef DSS_CAT_
