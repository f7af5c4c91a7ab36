// splitter_buckets.cpp -- go / no-go measurement for a splitter-based (sample sort) round 0 (round-5 verdict, item 1a):
// draw S sampled 64-bit round-0 keys, sort them, take 65 535 evenly spaced ones as order-preserving splitters, rank every
// suffix by splitter index and look at the bucket sizes: what share of the suffixes falls into buckets that one workgroup
// could finish inside LDS (<= 12 288 entries)?  Keys as the product makes them: coded (dq_alpha_code.h / dq_coded_keys.h)
// when the byte histogram allows, else the first 8 raw bytes big-endian.
//   g++ -O2 -fopenmp -std=c++17 -I deltaq_amd/csrc tools/exp/splitter_buckets.cpp -o /tmp/exp/splitter_buckets
//   /tmp/exp/splitter_buckets <file> [bytes] [offset] [nsplit=65536] [sample=1048576] [coded=auto|0|1]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "dq_alpha_code.h"
#include "dq_coded_keys.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    fseek(f, 0, SEEK_END);
    long fsz = ftell(f);
    long off = argc > 3 ? atol(argv[3]) : 0;
    long n = argc > 2 && atol(argv[2]) > 0 ? atol(argv[2]) : fsz - off;
    const int nsplit = argc > 4 ? atoi(argv[4]) : 65536;
    const long S = argc > 5 ? atol(argv[5]) : 1 << 20;
    const char *coded_arg = argc > 6 ? argv[6] : "auto";
    std::vector<uint8_t> T((size_t)n + 64, 0);
    fseek(f, off, SEEK_SET);
    if (fread(T.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    int64_t hist[256] = {0};
    for (long i = 0; i < n; ++i) hist[T[i]]++;
    double h0 = 0;
    for (int b = 0; b < 256; ++b) if (hist[b]) { double p = (double)hist[b] / n; h0 -= p * std::log2(p); }
    dq::AlphaCode code;
    bool coded = dq::build_alpha_code(hist, &code) && code.avg_len <= 5.8 && h0 <= 5.8 - 0.25;
    if (!strcmp(coded_arg, "0")) coded = false;
    if (!strcmp(coded_arg, "1")) coded = true;
    printf("n=%ld h0=%.3f sigma=%d avg_len=%.3f coded=%d nsplit=%d sample=%ld\n", n, h0, code.sigma, code.avg_len, (int)coded, nsplit, S);
    std::vector<uint64_t> K((size_t)n);
    const long n4 = (n + 3) / 4;
#pragma omp parallel for schedule(static)
    for (long q = 0; q < n4; ++q) {
        const long i0 = q * 4;
        if (coded) {
            uint32_t w[5];
            uint8_t tmp[20] = {0};
            const long avail = std::min<long>(20, n + 64 - i0);
            memcpy(tmp, &T[i0], (size_t)avail);
            memcpy(w, tmp, 20);
            uint64_t key[4];
            dq::coded_keys4(w, code.tab, key);
            for (int c = 0; c < 4 && i0 + c < n; ++c) K[i0 + c] = key[c];
        } else {
            for (int c = 0; c < 4 && i0 + c < n; ++c) {
                uint64_t k = 0;
                for (int b = 0; b < 8; ++b) k = (k << 8) | T[i0 + c + b];
                K[i0 + c] = k;
            }
        }
    }
    std::mt19937_64 rng(12345);
    std::vector<uint64_t> smp((size_t)S);
    for (long i = 0; i < S; ++i) smp[i] = K[rng() % (uint64_t)n];
    std::sort(smp.begin(), smp.end());
    std::vector<uint64_t> spl;
    for (int j = 1; j < nsplit; ++j) spl.push_back(smp[(size_t)((double)j * S / nsplit)]);
    spl.erase(std::unique(spl.begin(), spl.end()), spl.end());
    const int nb = (int)spl.size() + 1;
    std::vector<long> cnt((size_t)nb, 0);
#pragma omp parallel
    {
        std::vector<long> loc((size_t)nb, 0);
#pragma omp for schedule(static)
        for (long i = 0; i < n; ++i) loc[std::upper_bound(spl.begin(), spl.end(), K[i]) - spl.begin()]++;
#pragma omp critical
        for (int b = 0; b < nb; ++b) cnt[b] += loc[b];
    }
    // bucket b holds keys in (spl[b-1], spl[b]] ... a splitter value itself lands in the bucket right of it with
    // upper_bound; a heavy key (many equal keys) is one bucket's problem whatever the splitters do
    for (long cap : {4096L, 8192L, 12288L, 16384L, 32768L}) {
        long in = 0, big = 0;
        for (int b = 0; b < nb; ++b) { if (cnt[b] <= cap) in += cnt[b]; else ++big; }
        printf("cap %6ld: %.2f %% of the suffixes in buckets <= cap, %d buckets above it (of %d distinct)\n", cap, 100.0 * in / n, (int)big, nb);
    }
    std::vector<long> sorted_cnt(cnt);
    std::sort(sorted_cnt.begin(), sorted_cnt.end());
    printf("bucket sizes: median %ld, p90 %ld, p99 %ld, p99.9 %ld, max %ld\n", sorted_cnt[nb / 2], sorted_cnt[(size_t)(nb * 0.9)],
           sorted_cnt[(size_t)(nb * 0.99)], sorted_cnt[(size_t)(nb * 0.999)], sorted_cnt[nb - 1]);
    // the irreducible part: suffixes whose KEY VALUE alone occurs more than cap times (full sort of the keys)
    std::sort(K.begin(), K.end());
    long tied = 0, heavy12 = 0, heavy512 = 0, groups12 = 0;
    for (long i = 0; i < n;) {
        long j = i + 1;
        while (j < n && K[j] == K[i]) ++j;
        const long g = j - i;
        if (g > 1) tied += g;
        if (g > 12288) { heavy12 += g; ++groups12; }
        if (g > 512) heavy512 += g;
        i = j;
    }
    printf("equal-key groups: %.2f %% of the suffixes tied at all, %.2f %% in groups > 512, %.2f %% in %ld groups > 12288 (no splitter separates those)\n",
           100.0 * tied / n, 100.0 * heavy512 / n, 100.0 * heavy12 / n, groups12);
    return 0;
}
