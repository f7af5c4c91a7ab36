// bsort.hip -- developer micro-benchmark for bucket_sort_kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/bsort.hip -o tools/kbench/bsort
//   tools/kbench/bsort [log2_n=26] [X=1536]
// Input: n words (key36 << ib | suffix) with uniformly random keys, already grouped by their top 16 key bits
// (what the two digit passes of the bucketed round 0 leave).  Prints the kernel time and, with phase stamps
// compiled in, the average time a workgroup spends in each phase.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define DQ_BKT_PHASE_TIMING 1
#include "../../deltaq_amd/csrc/dq_bucket_sort.h"

using namespace dq;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void gen_kernel(uint64_t *w, int64_t n, int ib)
{
    // bucket b owns positions [b*n/65536, (b+1)*n/65536): equal-size buckets are close enough to Poisson ones
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        const uint64_t b = (uint64_t)((__int128)i * 65536 / n);
        const uint64_t key = (b << 20) | (x & 0xfffff);
        w[i] = (key << ib) | (uint64_t)i;
    }
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 26;
    const int64_t X = argc > 2 ? atoll(argv[2]) : 1536;
    const int64_t grid = argc > 3 ? atoll(argv[3]) : 256;
    const int64_t n = 1ll << lg;
    const int ib = lg, lowbits = 20;
    const int64_t C = kBktCap - X, ntiles = (n + C - 1) / C;
    uint64_t *W; int32_t *SA; uint32_t *ebits; int64_t *bounds; BucketFlags *flags; long long *ts;
    CK(hipMalloc(&W, n * 8)); CK(hipMalloc(&SA, n * 4)); CK(hipMalloc(&ebits, n / 8 + 64)); CK(hipMalloc(&bounds, (ntiles + 2) * 8));
    CK(hipMalloc(&flags, 64)); CK(hipMemset(flags, 0, 64)); CK(hipMemset(ebits, 0, n / 8 + 64));
    CK(hipMalloc(&ts, ntiles * 16 * 8)); CK(hipMemset(ts, 0, ntiles * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_bkt_ts), &ts, sizeof ts));
    hipLaunchKernelGGL(gen_kernel, dim3(2048), dim3(256), 0, 0, W, n, ib);
    hipLaunchKernelGGL(bucket_bounds_kernel, dim3((unsigned)((ntiles + 256) / 256)), dim3(256), 0, 0, (const uint64_t *)W, n, ib + lowbits, C, X, ntiles, bounds, flags);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(bucket_sort_kernel<int32_t>, dim3((unsigned)std::min<int64_t>(ntiles, grid)), dim3(kBktThreads), 0, 0, (const uint64_t *)W, ib, lowbits,
                           (const int64_t *)bounds, ntiles, SA, ebits, flags);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    CK(hipGetLastError());
    unsigned long long ov; CK(hipMemcpy(&ov, flags, 8, hipMemcpyDeviceToHost));
    printf("n=2^%d X=%lld C=%lld tiles=%lld : %.1f us  (%.1f G elements/s)%s\n", lg, (long long)X, (long long)C, (long long)ntiles, best * 1e3,
           n / (best * 1e-3) / 1e9, ov ? "  OVERFLOW" : "");
    std::vector<long long> h(ntiles * 16); CK(hipMemcpy(h.data(), ts, ntiles * 16 * 8, hipMemcpyDeviceToHost));
    double acc[16] = {0}; long long cnt = 0;
    for (int64_t t = 0; t < ntiles; ++t) { if (!h[t * 16]) continue; ++cnt; for (int i = 1; i < 16; ++i) if (h[t * 16 + i]) acc[i] += (double)(h[t * 16 + i] - h[t * 16 + i - 1]); }
    const char *nm[16] = {"", "edges + zero bins (words land)", "keys, next fetch, count atomics", "scan", "scatter", "bin walk", "suffix exchange", "store"};
    for (int i = 1; i < 8; ++i) printf("  phase %-32s avg %9.0f ticks\n", nm[i], cnt ? acc[i] / cnt : 0.0);
    // check order on a sample
    std::vector<uint64_t> hw(n); std::vector<int32_t> hs(n);
    CK(hipMemcpy(hw.data(), W, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), SA, n * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    for (int64_t i = 1; i < n; ++i) if ((hw[hs[i - 1]] >> ib) > (hw[hs[i]] >> ib)) ++bad;      // suffix index == word position here
    printf("  order violations: %lld\n", bad);
    return 0;
}
