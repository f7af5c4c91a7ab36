/*
 * checkers_impl.h -- body of the checkers, instantiated for int32_t / int64_t
 * indices by checkers.c (define IDX and SUF before including).
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* ReadOnlySpan<byte>.SequenceCompareTo(other): lexicographic over unsigned
 * bytes, a proper prefix sorts first.  Used by Verify at
 * LibDivSufSortTests.cs:50-52. */
static int FN(seqcmp)(const uint8_t *T, int64_t n, int64_t a, int64_t b)
{
    int64_t la = n - a, lb = n - b;
    int64_t m = la < lb ? la : lb;
    int c = memcmp(T + a, T + b, (size_t)m);
    if (c != 0) return c;
    return (la > lb) - (la < lb);
}

/* LibDivSufSortTests.cs:43-59: for i in [0, n-1): require suffix(SA[i]) < suffix(SA[i+1]). */
int64_t FN(dq_oracle_verify_strict)(const uint8_t *T, const IDX *SA, int64_t n)
{
    for (int64_t i = 0; i + 1 < n; ++i) {
        if (!(FN(seqcmp)(T, n, (int64_t)SA[i], (int64_t)SA[i + 1]) < 0)) return i;
    }
    return -1;
}

int64_t FN(dq_oracle_verify_sampled)(const uint8_t *T, const IDX *SA, int64_t n,
                                     int64_t samples, uint64_t seed)
{
    if (n < 2) return -1;
    uint64_t x = seed;
    for (int64_t s = 0; s < samples; ++s) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        int64_t i = (int64_t)(z % (uint64_t)(n - 1));
        if (!(FN(seqcmp)(T, n, (int64_t)SA[i], (int64_t)SA[i + 1]) < 0)) return i;
    }
    return -1;
}

/* LDSSChecker.Check, LDSSChecker.cs:23-119 (sufcheck). */
int32_t FN(dq_oracle_sufcheck)(const uint8_t *T, int64_t n, const IDX *SA, int64_t sa_len)
{
    /* :29-33 argument check */
    if (n != sa_len) return DQ_CHECK_BAD_ARGUMENTS;
    /* :35-39 empty text is Done */
    if (n == 0) return DQ_CHECK_DONE;

    /* :43-55 range check */
    for (int64_t i = 0; i < n; ++i) {
        if (SA[i] < 0 || (IDX)n <= SA[i]) return DQ_CHECK_OUT_OF_RANGE;
    }
    /* :58-70 first characters must be non-decreasing */
    for (int64_t i = 1; i < n; ++i) {
        if (T[SA[i - 1]] > T[SA[i]]) return DQ_CHECK_WRONG_ORDER;
    }
    /* :73-84 bucket starts C[c] */
    IDX C[256];
    memset(C, 0, sizeof C);
    for (int64_t i = 0; i < n; ++i) ++C[T[i]];
    IDX p = 0;
    for (int c = 0; c < 256; ++c) { IDX t = C[c]; C[c] = p; p += t; }

    /* :86-114 walk: the suffix preceding SA[i] must sit at the next free slot
     * of its first character's bucket. */
    IDX q = C[T[n - 1]];
    C[T[n - 1]] += 1;
    for (int64_t i = 0; i < n; ++i) {
        IDX t;
        int c;
        p = SA[i];
        if (0 < p) {
            c = T[--p];
            t = C[c];
        } else {
            c = T[p = (IDX)(n - 1)];
            t = q;
        }
        if (t < 0 || p != SA[t]) return DQ_CHECK_WRONG_POSITION;
        if (t != q) {
            ++C[c];
            if ((IDX)n <= C[c] || T[SA[C[c]]] != c) C[c] = -1;
        }
    }
    return DQ_CHECK_DONE;
}

#undef FN
#undef CAT
#undef CAT_
