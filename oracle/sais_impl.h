/*
 * sais_impl.h -- body of the SAIS restatement, instantiated by sais.c
 *   IDX     index type (int32_t / int64_t)
 *   CHR     text element type of THIS level (uint8_t at the top, IDX in the recursion)
 *   LVL     name suffix of this level's functions   (e.g. _u8_i32)
 *   REC     name suffix of the recursion's functions (always the IDX-text instantiation)
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h).
 *
 * Restates, function by function, the reference's second suffix sorting provider
 *   src/DeltaQ.SuffixSorting.SAIS/SAIS.cs   (Yuta Mori's sais-lite 2.4.1 in C#)
 * so that results can be bit-compared against a second, linear-time CPU implementation that is independent of
 * the LibDivSufSort restatement.
 * TextAccessor<T>.this[int] (TextAccessor.cs:12-16) is plain indexing here.  `c == b` on Span<int>
 * (same memory) is pointer equality.  `new int[k]` is calloc (C# zero-fills).
 */

#define SCAT_(a, b) a##b
#define SCAT(a, b) SCAT_(a, b)
#define SFN(name) SCAT(name, LVL)
#define SREC(name) SCAT(name, REC)

/* SAIS.cs:51-60 */
static void SFN(sais_get_counts)(const CHR *T, IDX *c, IDX n, IDX k)
{
    for (IDX i = 0; i < k; ++i) c[i] = 0;
    for (IDX i = 0; i < n; ++i) c[T[i]]++;
}

/* SAIS.cs:62-70 */
static void SFN(sais_get_buckets)(const IDX *c, IDX *b, IDX k, int end)
{
    IDX sum = 0;
    for (IDX i = 0; i < k; ++i) {
        sum += c[i];
        b[i] = end ? sum : sum - c[i];
    }
}

/* SAIS.cs:75-134  sort all type LMS suffixes */
static void SFN(sais_lms_sort)(const CHR *T, IDX *sa, IDX *c, IDX *b, IDX n, IDX k)
{
    IDX bb, i, j;
    IDX c0, c1;

    /* compute SAl  (:80-108) */
    if (c == b) SFN(sais_get_counts)(T, c, n, k);
    SFN(sais_get_buckets)(c, b, k, 0);          /* find starts of buckets */

    j = n - 1;
    bb = b[c1 = (IDX)T[j]];
    --j;
    sa[bb++] = (IDX)T[j] < c1 ? ~j : j;
    for (i = 0; i < n; ++i) {
        if (0 < (j = sa[i])) {
            if ((c0 = (IDX)T[j]) != c1) {
                b[c1] = bb;
                bb = b[c1 = c0];
            }
            --j;
            sa[bb++] = (IDX)T[j] < c1 ? ~j : j;
            sa[i] = 0;
        } else if (j < 0) {
            sa[i] = ~j;
        }
    }

    /* compute SAs  (:110-133) */
    if (c == b) SFN(sais_get_counts)(T, c, n, k);
    SFN(sais_get_buckets)(c, b, k, 1);          /* find ends of buckets */

    for (i = n - 1, bb = b[c1 = 0]; 0 <= i; --i) {
        if (0 < (j = sa[i])) {
            if ((c0 = (IDX)T[j]) != c1) {
                b[c1] = bb;
                bb = b[c1 = c0];
            }
            --j;
            sa[--bb] = (IDX)T[j] > c1 ? ~(j + 1) : j;
            sa[i] = 0;
        }
    }
}

/* SAIS.cs:136-215 */
static IDX SFN(sais_lms_post_proc)(const CHR *T, IDX *sa, IDX n, IDX m)
{
    IDX i, j, p, q;
    IDX qlen, name;
    IDX c0, c1;

    /* compact all the sorted substrings into the first m items of SA; 2*m must be not larger than n  (:142-163) */
    for (i = 0; (p = sa[i]) < 0; ++i) sa[i] = ~p;
    if (i < m) {
        for (j = i, ++i;; ++i) {
            if ((p = sa[i]) < 0) {
                sa[j++] = ~p;
                sa[i] = 0;
                if (j == m) break;
            }
        }
    }

    /* store the length of all substrings  (:165-187) */
    i = n - 1;
    j = n - 1;
    c0 = (IDX)T[n - 1];
    do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) >= c1);
    for (; 0 <= i;) {
        do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) <= c1);
        if (0 <= i) {
            sa[m + ((i + 1) >> 1)] = j - i;
            j = i + 1;
            do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) >= c1);
        }
    }

    /* find the lexicographic names of all substrings  (:189-212) */
    for (i = 0, name = 0, q = n, qlen = 0; i < m; ++i) {
        p = sa[i];
        IDX plen = sa[m + (p >> 1)];
        int diff = 1;
        if (plen == qlen && q + plen < n) {
            for (j = 0; j < plen && T[p + j] == T[q + j]; ++j) { }
            if (j == plen) diff = 0;
        }
        if (diff) {
            ++name;
            q = p;
            qlen = plen;
        }
        sa[m + (p >> 1)] = name;
    }
    return name;
}

/* SAIS.cs:217-273 */
static void SFN(sais_induce_sa)(const CHR *T, IDX *sa, IDX *c, IDX *b, IDX n, IDX k)
{
    IDX bb, i, j;
    IDX c0, c1;

    /* compute SAl  (:222-245) */
    if (c == b) SFN(sais_get_counts)(T, c, n, k);
    SFN(sais_get_buckets)(c, b, k, 0);          /* find starts of buckets */

    j = n - 1;
    bb = b[c1 = (IDX)T[j]];
    sa[bb++] = (0 < j && (IDX)T[j - 1] < c1) ? ~j : j;
    for (i = 0; i < n; ++i) {
        j = sa[i];
        sa[i] = ~j;
        if (0 < j) {
            if ((c0 = (IDX)T[--j]) != c1) {
                b[c1] = bb;
                bb = b[c1 = c0];
            }
            sa[bb++] = (0 < j && (IDX)T[j - 1] < c1) ? ~j : j;
        }
    }

    /* compute SAs  (:247-272) */
    if (c == b) SFN(sais_get_counts)(T, c, n, k);
    SFN(sais_get_buckets)(c, b, k, 1);          /* find ends of buckets */

    for (i = n - 1, bb = b[c1 = 0]; 0 <= i; --i) {
        if (0 < (j = sa[i])) {
            if ((c0 = (IDX)T[--j]) != c1) {
                b[c1] = bb;
                bb = b[c1 = c0];
            }
            sa[--bb] = (j == 0 || (IDX)T[j - 1] > c1) ? ~j : j;
        } else {
            sa[i] = ~j;
        }
    }
}

#ifndef SAIS_MIN_BUCKET_SIZE
#define SAIS_MIN_BUCKET_SIZE 256             /* SAIS.cs:48  MinBucketSize = byte.MaxValue + 1 */
#endif

/* SAIS.cs:279-494  find the suffix array SA of T[0..n-1] in {0..k-1}^n; sa has n + fs usable entries.
 * Returns 0, or -2 when an allocation fails. */
static int SFN(sais_main)(const CHR *T, IDX *sa, IDX fs, IDX n, IDX k)
{
    IDX *c, *b;
    IDX *c_own = 0, *b_own = 0;              /* what this level allocated itself (`new int[k]`) */
    IDX i, j, bb, m;
    IDX name;
    IDX c0, c1;
    unsigned flags;
    int rc = 0;

    /* :287-325 where the bucket arrays live */
    if (k <= SAIS_MIN_BUCKET_SIZE) {
        c = c_own = (IDX *)calloc((size_t)k, sizeof(IDX));
        if (!c) return -2;
        if (k <= fs) {
            b = sa + (n + fs - k);
            flags = 1;
        } else {
            b = b_own = (IDX *)calloc((size_t)k, sizeof(IDX));
            if (!b) { free(c_own); return -2; }
            flags = 3;
        }
    } else if (k <= fs) {
        c = sa + (n + fs - k);
        if (k <= fs - k) {
            b = sa + (n + fs - k * 2);
            flags = 0;
        } else if (k <= SAIS_MIN_BUCKET_SIZE * 4) {
            b = b_own = (IDX *)calloc((size_t)k, sizeof(IDX));
            if (!b) return -2;
            flags = 2;
        } else {
            b = c;
            flags = 8;
        }
    } else {
        c = b = c_own = (IDX *)calloc((size_t)k, sizeof(IDX));
        if (!c) return -2;
        flags = 4 | 8;
    }

    /* stage 1: reduce the problem by at least 1/2; sort all the LMS-substrings  (:327-377) */
    SFN(sais_get_counts)(T, c, n, k);
    SFN(sais_get_buckets)(c, b, k, 1);          /* find ends of buckets */

    for (i = 0; i < n; ++i) sa[i] = 0;

    bb = -1;
    i = n - 1;
    j = n;
    m = 0;
    c0 = (IDX)T[n - 1];
    do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) >= c1);
    for (; 0 <= i;) {
        do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) <= c1);
        if (0 <= i) {
            if (0 <= bb) sa[bb] = j;
            bb = --b[c1];
            j = i;
            ++m;
            do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) >= c1);
        }
    }
    if (1 < m) {
        SFN(sais_lms_sort)(T, sa, c, b, n, k);
        name = SFN(sais_lms_post_proc)(T, sa, n, m);
    } else if (m == 1) {
        sa[bb] = j + 1;
        name = 1;
    } else {
        name = 0;
    }

    /* stage 2: solve the reduced problem; recurse if names are not yet unique  (:379-455) */
    if (name < m) {
        if (flags & 4) { free(c_own); c_own = 0; c = 0; b = 0; }
        if (flags & 2) { free(b_own); b_own = 0; b = 0; }
        IDX newfs = n + fs - m * 2;
        if ((flags & (1 | 4 | 8)) == 0) {
            if (k + name <= newfs) newfs -= k;
            else flags |= 8;
        }

        for (i = m + (n >> 1) - 1, j = m * 2 + newfs - 1; m <= i; --i) {
            if (sa[i] != 0) sa[j--] = sa[i] - 1;
        }

        rc = SREC(sais_main)(sa + (m + newfs), sa, newfs, m, name);
        if (rc != 0) { free(c_own); free(b_own); return rc; }

        i = n - 1;
        j = m * 2 - 1;
        c0 = (IDX)T[n - 1];
        do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) >= c1);
        for (; 0 <= i;) {
            do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) <= c1);
            if (0 <= i) {
                sa[j--] = i + 1;
                do { c1 = c0; } while (0 <= --i && (c0 = (IDX)T[i]) >= c1);
            }
        }

        for (i = 0; i < m; ++i) sa[i] = sa[m + sa[i]];
        if (flags & 4) {
            c = b = c_own = (IDX *)calloc((size_t)k, sizeof(IDX));
            if (!c) { free(b_own); return -2; }
        }
        if (flags & 2) {
            b = b_own = (IDX *)calloc((size_t)k, sizeof(IDX));
            if (!b) { free(c_own); return -2; }
        }
    }

    /* stage 3: induce the result for the original problem  (:457-493) */
    if (flags & 8) SFN(sais_get_counts)(T, c, n, k);
    /* put all left-most S characters into their buckets */
    if (1 < m) {
        SFN(sais_get_buckets)(c, b, k, 1);      /* find ends of buckets */
        i = m - 1;
        j = n;
        IDX p = sa[m - 1];
        c1 = (IDX)T[p];
        do {
            IDX q = b[c0 = c1];
            while (q < j) sa[--j] = 0;
            do {
                sa[--j] = p;
                if (--i < 0) break;
                p = sa[i];
            } while ((c1 = (IDX)T[p]) == c0);
        } while (0 <= i);
        while (0 < j) sa[--j] = 0;
    }

    SFN(sais_induce_sa)(T, sa, c, b, n, k);
    free(c_own);
    free(b_own);
    return 0;
}

#undef SFN
#undef SREC
#undef SCAT
#undef SCAT_
