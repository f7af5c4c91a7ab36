/*
 * gen.c -- synthetic workload generators shared by tests and bench.py
 * (BASELINE.json configs).  Integer-only so every platform
 * produces identical bytes.  Data generation only: not part of the sort path and
 * not part of the oracle.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void dq_gen_uniform(uint8_t *out, int64_t n, uint64_t seed);
void dq_gen_enwik_like(uint8_t *out, int64_t n, uint64_t seed, int64_t repeat_period);

typedef struct { uint64_t x; } sm64_t;

static inline uint64_t sm64_next(sm64_t *s)
{
    s->x += 0x9E3779B97F4A7C15ull;
    uint64_t z = s->x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t sm64_below(sm64_t *s, uint64_t m) { return sm64_next(s) % m; }

/* i.i.d. uniform bytes: 8 bytes per draw, little-endian. */
void dq_gen_uniform(uint8_t *out, int64_t n, uint64_t seed)
{
    sm64_t s = { seed };
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t z = sm64_next(&s);
        for (int k = 0; k < 8; ++k) out[i + k] = (uint8_t)(z >> (8 * k));
    }
    if (i < n) {
        uint64_t z = sm64_next(&s);
        for (int k = 0; i < n; ++i, ++k) out[i] = (uint8_t)(z >> (8 * k));
    }
}

/* ---- enwik8-style skewed text: Zipf word model + XML boilerplate + repeats ---- */
#define VOCAB 50000

static const char LETTERS[26] = {'e','t','a','o','i','n','s','h','r','d','l','c','u',
                                 'm','w','f','g','y','p','b','v','k','j','x','q','z'};
static const uint32_t LETTER_W[26] = {1200,900,800,750,700,670,630,600,590,430,400,280,275,
                                      240,236,222,201,197,190,150,98,77,15,15,10,7};

typedef struct {
    uint8_t *out; int64_t n, pos;
    int col, next_break;
} sink_t;

static inline void put(sink_t *k, uint8_t c)
{
    if (k->pos < k->n) k->out[k->pos] = c;
    k->pos++;
}
static void puts_(sink_t *k, const char *s) { while (*s) put(k, (uint8_t)*s++); }

void dq_gen_enwik_like(uint8_t *out, int64_t n, uint64_t seed, int64_t R)
{
    sm64_t s = { seed };
    if (R <= 0) R = 256 * 1024;

    /* vocabulary */
    uint32_t lcum[26], ltot = 0;
    for (int i = 0; i < 26; ++i) { ltot += LETTER_W[i]; lcum[i] = ltot; }
    char (*words)[16] = malloc((size_t)VOCAB * 16);
    uint8_t *wlen = malloc(VOCAB);
    uint64_t *zcum = malloc((size_t)VOCAB * sizeof(uint64_t));
    uint64_t ztot = 0;
    for (int w = 0; w < VOCAB; ++w) {
        int len = 1 + (int)sm64_below(&s, 3) + (int)sm64_below(&s, 4) + (int)sm64_below(&s, 5);
        wlen[w] = (uint8_t)len;
        for (int j = 0; j < len; ++j) {
            uint32_t r = (uint32_t)sm64_below(&s, ltot);
            int li = 0;
            while (lcum[li] <= r) ++li;
            words[w][j] = LETTERS[li];
        }
        words[w][len] = 0;
        ztot += (1ull << 32) / (uint64_t)(w + 1);
        zcum[w] = ztot;
    }

    sink_t k = { out, n, 0, 0, 72 + (int)sm64_below(&s, 16) };
    int64_t next_page = 0;
    int64_t next_copy = R / 2 + (int64_t)sm64_below(&s, (uint64_t)R);
    int64_t page_id = 1;

    while (k.pos < n) {
        if (k.pos >= next_page) {
            char buf[32];
            puts_(&k, "\n  <page>\n    <title>");
            for (int t = 0; t < 2; ++t) {
                uint64_t r = sm64_below(&s, ztot);
                int lo = 0, hi = VOCAB - 1;
                while (lo < hi) { int mid = (lo + hi) >> 1; if (zcum[mid] <= r) lo = mid + 1; else hi = mid; }
                if (t) put(&k, ' ');
                puts_(&k, words[lo]);
            }
            puts_(&k, "</title>\n    <id>");
            { /* decimal page id */
                int64_t v = page_id++; int bl = 0; char tmp[24];
                do { tmp[bl++] = (char)('0' + v % 10); v /= 10; } while (v);
                for (int j = 0; j < bl; ++j) buf[j] = tmp[bl - 1 - j];
                buf[bl] = 0;
            }
            puts_(&k, buf);
            puts_(&k, "</id>\n    <revision>\n      <text xml:space=\"preserve\">");
            k.col = 0;
            next_page = k.pos + 1024 + (int64_t)sm64_below(&s, 6144);
            continue;
        }
        if (k.pos >= next_copy && k.pos > 1024) {
            int e = 8 + (int)sm64_below(&s, 9);
            int64_t len = ((int64_t)1 << e) + (int64_t)sm64_below(&s, (uint64_t)1 << e);
            if (len > 65536) len = 65536;
            if (len > k.pos) len = k.pos;
            int64_t src = (int64_t)sm64_below(&s, (uint64_t)(k.pos - len + 1));
            for (int64_t j = 0; j < len; ++j) {
                uint8_t c = (src + j < n) ? out[src + j] : (uint8_t)' ';
                put(&k, c);
            }
            next_copy = k.pos + R / 2 + (int64_t)sm64_below(&s, (uint64_t)R);
            continue;
        }
        /* one word */
        uint64_t r = sm64_below(&s, ztot);
        int lo = 0, hi = VOCAB - 1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (zcum[mid] <= r) lo = mid + 1; else hi = mid; }
        uint64_t style = sm64_below(&s, 100);
        int start = (int)k.pos;
        (void)start;
        if (style < 3) {
            puts_(&k, "[["); puts_(&k, words[lo]); puts_(&k, "]]");
            k.col += wlen[lo] + 4;
        } else if (style < 4) {
            put(&k, (uint8_t)(words[lo][0] - 'a' + 'A'));
            puts_(&k, words[lo] + 1);
            put(&k, '.');
            k.col += wlen[lo] + 1;
        } else {
            puts_(&k, words[lo]);
            k.col += wlen[lo];
        }
        if (k.col >= k.next_break) {
            put(&k, '\n');
            k.col = 0;
            k.next_break = 72 + (int)sm64_below(&s, 16);
        } else {
            put(&k, ' ');
            k.col += 1;
        }
    }
    free(words); free(wlen); free(zcum);
}
