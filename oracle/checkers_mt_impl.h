/*
 * checkers_mt_impl.h -- LDSSChecker.Check (LDSSChecker.cs:23-119) evaluated by several threads,
 * instantiated for int32_t / int64_t indices by checkers_mt.c (define IDX and SUF first).
 * TEST INFRASTRUCTURE ONLY (see dq_oracle.h).
 *
 * Same phases, same order, same result code as the sequential restatement in checkers_impl.h
 * (which stays the pinned one; tests/test_oracle_golden.py compares the two on correct and on
 * corrupted arrays):
 *   :43-55  range check                       -- independent per entry
 *   :58-70  first characters non-decreasing   -- independent per adjacent pair
 *   :73-84  bucket starts                     -- byte histogram of T, summed over threads
 *   :86-114 the walk "the suffix preceding SA[i] sits at the next free slot of its first
 *           character's bucket".  The only state carried along i is C[c], and C[c] moves by one
 *           per entry whose preceding character is c; so a thread that owns [i0, i1) can start
 *           from C[c] + (number of such entries before i0), which a counting pass provides.
 *           The reference's "C[c] = -1 once the bucket is exhausted" is kept per thread: when it
 *           fires on a thread's last entry for c, the next thread starts at that same exhausted
 *           slot t (t >= n, or T[SA[t]] != c) and fails its own position test there, because
 *           T[p] == c != T[SA[t]] -- the verdict is the same WrongPosition.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

typedef struct {
    const uint8_t *T;
    const IDX *SA;
    int64_t n, i0, i1;
    int phase;
    int64_t hist[256];      /* phase 1: bytes of T in [i0, i1); phase 2: preceding characters of SA[i0..i1) */
    int64_t start[256];     /* phase 3: C[] at i0 */
    int64_t q;
    int fail;
} FN(mt_job);

static void *FN(mt_worker)(void *arg)
{
    FN(mt_job) *j = (FN(mt_job) *)arg;
    const uint8_t *T = j->T;
    const IDX *SA = j->SA;
    const int64_t n = j->n;
    j->fail = 0;
    if (j->phase == 0) {                    /* range */
        for (int64_t i = j->i0; i < j->i1; ++i)
            if (SA[i] < 0 || (IDX)n <= SA[i]) { j->fail = 1; return 0; }
    } else if (j->phase == 1) {             /* first characters + byte histogram of T */
        memset(j->hist, 0, sizeof j->hist);
        for (int64_t i = j->i0; i < j->i1; ++i) {
            if (i > 0 && T[SA[i - 1]] > T[SA[i]]) { j->fail = 1; return 0; }
            ++j->hist[T[i]];
        }
    } else if (j->phase == 2) {             /* how far each C[c] moves inside [i0, i1) */
        memset(j->hist, 0, sizeof j->hist);
        for (int64_t i = j->i0; i < j->i1; ++i) {
            const int64_t p = (int64_t)SA[i];
            if (p > 0) ++j->hist[T[p - 1]];
        }
    } else {                                /* the walk, from this thread's own C[] */
        int64_t C[256];
        memcpy(C, j->start, sizeof C);
        const int64_t q = j->q;
        for (int64_t i = j->i0; i < j->i1; ++i) {
            int64_t p = (int64_t)SA[i], t;
            int c;
            if (0 < p) { c = T[--p]; t = C[c]; }
            else { c = T[p = n - 1]; t = q; }
            if (t < 0 || t >= n || p != (int64_t)SA[t]) { j->fail = 1; return 0; }
            if (t != q) {
                ++C[c];
                if (n <= C[c] || T[SA[C[c]]] != c) C[c] = -1;
            }
        }
    }
    return 0;
}

int32_t FN(dq_oracle_sufcheck_mt)(const uint8_t *T, int64_t n, const IDX *SA, int64_t sa_len, int32_t threads)
{
    if (n != sa_len) return DQ_CHECK_BAD_ARGUMENTS;
    if (n == 0) return DQ_CHECK_DONE;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if ((int64_t)threads > n) threads = (int32_t)n;
    FN(mt_job) *jobs = (FN(mt_job) *)calloc((size_t)threads, sizeof *jobs);
    pthread_t *tid = (pthread_t *)calloc((size_t)threads, sizeof *tid);
    char *started = (char *)calloc((size_t)threads, 1);
    if (!jobs || !tid || !started) { free(jobs); free(tid); free(started); return DQ_CHECK_BAD_ARGUMENTS; }
    for (int k = 0; k < threads; ++k) {
        jobs[k].T = T; jobs[k].SA = SA; jobs[k].n = n;
        jobs[k].i0 = n * k / threads;
        jobs[k].i1 = n * (k + 1) / threads;
    }
    int32_t rc = DQ_CHECK_DONE;
    int64_t C[256];
    static const int32_t codes[4] = {DQ_CHECK_OUT_OF_RANGE, DQ_CHECK_WRONG_ORDER, 0, DQ_CHECK_WRONG_POSITION};
    for (int phase = 0; phase < 4 && rc == DQ_CHECK_DONE; ++phase) {
        if (phase == 3) {
            /* :73-84 bucket starts, :86-88 the slot of suffix n-1, then each thread's starting C[] */
            /* (C holds the byte counts of T, summed after phase 1; the jobs' hist now holds phase 2's) */
            int64_t p = 0;
            for (int c = 0; c < 256; ++c) { const int64_t t = C[c]; C[c] = p; p += t; }
            const int64_t q = C[T[n - 1]];
            C[T[n - 1]] += 1;
            for (int k = 0; k < threads; ++k) {
                memcpy(jobs[k].start, C, sizeof C);
                jobs[k].q = q;
                for (int c = 0; c < 256; ++c) C[c] += jobs[k].hist[c];
            }
        }
        for (int k = 0; k < threads; ++k) {
            jobs[k].phase = phase;
            started[k] = pthread_create(&tid[k], 0, FN(mt_worker), &jobs[k]) == 0;
            if (!started[k]) FN(mt_worker)(&jobs[k]);             /* no thread to be had: do the share here */
        }
        for (int k = 0; k < threads; ++k)
            if (started[k]) pthread_join(tid[k], 0);
        for (int k = 0; k < threads; ++k)
            if (jobs[k].fail) rc = codes[phase];
        if (phase == 1) {
            memset(C, 0, sizeof C);
            for (int k = 0; k < threads; ++k)
                for (int c = 0; c < 256; ++c) C[c] += jobs[k].hist[c];
        }
    }
    free(jobs);
    free(tid);
    free(started);
    return rc;
}

#undef FN
#undef CAT
#undef CAT_
