// kbench.hip -- developer micro-benchmark for the radix kernels (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kbench/kbench.hip -o tools/kbench/kbench
//   tools/kbench/kbench [log2_n=26] [skew=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
// per-tile phase timestamps from thread 0 of radix_rank_kernel (the library's DQ_PHASE is empty)
__device__ long long *g_phase_ts = nullptr;          // [ntiles][8]
#define DQ_PHASE(i) do { if (threadIdx.x == 0 && g_phase_ts) g_phase_ts[(long long)s_tile * 8 + (i)] = clock64(); } while (0)
#include "../../deltaq_amd/csrc/dq_onesweep.h"

using namespace dq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void gen_kernel(uint64_t *k, int32_t *v, int64_t m, int skew)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x1234567;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        if (skew == 1) x &= 0x0303030303030303ull;            // 4 symbols per digit
        if (skew == 2) x = (x & 0xff) < 200 ? (x & ~0xffull) | 0x65 : x;   // 78% one digit value
        if (skew == 3) x = 0;                                  // all equal
        k[i] = x;
        v[i] = (int32_t)i;
    }
}

template <class F>
float time_it(F &&f, int reps = 5)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms); sum += ms;
    }
    CK(hipGetLastError());
    return sum / reps;
}

struct Bufs {
    uint64_t *k0, *k1, *k2; int32_t *v0, *v1, *v2;
    uint32_t *blockhist; int32_t *blockbase; uint32_t *partial; int64_t *digit_offset;
    uint32_t *status; OnesweepCtl *ctl; int64_t *sticky; uint8_t *text; int64_t *digit_offset_text; int64_t *bytehist;
};

template <int kItems, int kMinWaves, int kThreads = 256, bool kEarly = false, bool kLds = true, int kRounds = 1, int kPairMode = kPairs>
float run_onesweep(Bufs &B, int64_t m, int shift, bool synth, const char *tag)
{
    const int tile = kThreads * kItems;
    const int64_t ntiles = (m + tile - 1) / tile;
    auto f = [&]() {
        CK(hipMemsetAsync(B.status, 0, (size_t)ntiles * 256 * 4));
        CK(hipMemsetAsync(B.ctl, 0, sizeof(OnesweepCtl)));
        if constexpr (kItems % 4 == 0) { if (synth)
            hipLaunchKernelGGL((radix_rank_kernel<int32_t, uint32_t, kItems, kText, kMinWaves, kThreads, kEarly, kLds, kRounds>), dim3((unsigned)ntiles), dim3(kThreads), 0, 0,
                               (const uint64_t *)B.text, (const int32_t *)nullptr, B.k2, B.v2, m, 0, 64, 0, B.digit_offset_text, B.status, B.ctl, B.sticky); }
        if (!synth)
            hipLaunchKernelGGL((radix_rank_kernel<int32_t, uint32_t, kItems, kPairMode, kMinWaves, kThreads, kEarly, kLds, kRounds>), dim3((unsigned)ntiles), dim3(kThreads), 0, 0,
                               B.k0, (const int32_t *)B.v0, B.k2, B.v2, m, shift, 64, 0, B.digit_offset + (shift / 8) * 256, B.status, B.ctl, B.sticky);
    };
    float ms = time_it(f);
    OnesweepCtl h; CK(hipMemcpy(&h, B.ctl, sizeof h, hipMemcpyDeviceToHost));
    printf("%-22s thr=%4d items=%2d minw=%d ldsmatch=%d rounds=%d keysonly=%d : %8.1f us  %7.1f GB/s alg%s\n", tag, kThreads, kItems, kMinWaves, (int)kLds, kRounds, (int)(kPairMode == kKeys), ms * 1e3,
           (double)m * (synth ? 20 : (kPairMode == kKeys ? 16 : 24)) / (ms * 1e-3) / 1e9, h.error ? "  LOOKBACK TIMEOUT" : "");
    return ms;
}

bool same(const void *a, const void *b, size_t bytes)
{
    std::vector<char> ha(bytes), hb(bytes);
    CK(hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost));
    return memcmp(ha.data(), hb.data(), bytes) == 0;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int lg = argc > 1 ? atoi(argv[1]) : 26;
    const int skew = argc > 2 ? atoi(argv[2]) : 0;
    const int64_t m = ((int64_t)1 << lg) + (argc > 3 ? atoi(argv[3]) : 0);
    printf("m = %lld  skew=%d\n", (long long)m, skew);
    Bufs B;
    CK(hipMalloc(&B.k0, m * 8)); CK(hipMalloc(&B.k1, m * 8)); CK(hipMalloc(&B.k2, (m + (4 << 20)) * 8));
    CK(hipMalloc(&B.v0, m * 4)); CK(hipMalloc(&B.v1, m * 4)); CK(hipMalloc(&B.v2, (m + (4 << 20)) * 4));
    CK(hipMalloc(&B.partial, kHistBlocks * kMaxPasses * 256 * 4)); CK(hipMalloc(&B.digit_offset, kMaxPasses * 256 * 8));
    CK(hipMalloc(&B.status, ((size_t)m / 2048 + 2) * 256 * 4)); CK(hipMalloc(&B.ctl, sizeof(OnesweepCtl))); CK(hipMalloc(&B.sticky, 64)); CK(hipMalloc(&B.text, m + 64)); CK(hipMemset(B.text, 0, m + 64)); CK(hipMalloc(&B.digit_offset_text, 8 * 256 * 8)); CK(hipMalloc(&B.bytehist, 256 * 8));
    hipLaunchKernelGGL(gen_kernel, dim3(2048), dim3(256), 0, 0, B.k0, B.v0, m, skew);
    CK(hipMemcpy(B.text, B.k0, m, hipMemcpyDeviceToDevice));   // random bytes as text
    CK(hipDeviceSynchronize());
    {
        float t_th = time_it([&]() {
            CK(hipMemsetAsync(B.bytehist, 0, 256 * 8));
            hipLaunchKernelGGL(text_hist_kernel, dim3(kHistBlocks), dim3(kBlock), 0, 0, (const uint8_t *)B.text, m, (unsigned long long *)B.bytehist);
            hipLaunchKernelGGL(text_digit_offsets_kernel, dim3(8), dim3(kBlock), 0, 0, (const int64_t *)B.bytehist, (const uint8_t *)B.text, m, 8, B.digit_offset_text); });
        printf("text hist + reduce + offsets %8.1f us\n", t_th * 1e3);
    }

    const int shift = 8;
    // ---- global histograms
    float t_h = time_it([&]() { CK(hipMemsetAsync(B.partial, 0, 8 * 256 * 8));
                                hipLaunchKernelGGL(radix_hist_kernel<8>, dim3(kHistBlocks), dim3(kHistThreads), 0, 0, B.k0, m, reinterpret_cast<unsigned long long *>(B.partial), 0); });
    float t_hs = time_it([&]() { hipLaunchKernelGGL(radix_hist_scan_kernel, dim3(8), dim3(kHistScanThreads), 0, 0, reinterpret_cast<const unsigned long long *>(B.partial), B.digit_offset); });
    printf("hist (8 digits, one read) %8.1f us (%.1f GB/s)   hist_scan %6.1f us\n", t_h * 1e3, (double)m * 8 / (t_h * 1e-3) / 1e9, t_hs * 1e3);

    if (argc > 4) {     // profiling target: one variant only
        if (argv[4][0] == 'p') {      // phase timing
            const int64_t nt = (m + 12287) / 12288;
            long long *ts; CK(hipMalloc(&ts, nt * 8 * 8)); CK(hipMemset(ts, 0, nt * 8 * 8));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_ts), &ts, sizeof ts));
            run_onesweep<24, 2, 512, false, true, 2, kKeys>(B, m, shift, false, "keys (phase-timed)");
            std::vector<long long> h(nt * 8); CK(hipMemcpy(h.data(), ts, nt * 8 * 8, hipMemcpyDeviceToHost));
            double acc[8] = {0}; long long cnt = 0;
            for (int64_t t = 0; t < nt - 1; ++t) { bool ok = true; for (int i = 0; i < 8; ++i) ok &= h[t * 8 + i] != 0; if (!ok) continue; ++cnt;
                for (int i = 1; i < 8; ++i) acc[i] += (double)(h[t * 8 + i] - h[t * 8 + i - 1]); acc[0] += (double)(h[t * 8 + 7] - h[t * 8]); }
            const char *nm[8] = {"total", "ticket + load keys", "rank loop", "barrier", "digit scan + publish", "LDS exchange (all rounds)", "look-back resolve", "store issue"};
            for (int i = 0; i < 8; ++i) printf("  phase %-26s avg %10.0f ticks\n", nm[i], acc[i] / cnt);
            long long tmin = h[0], tmax = 0; for (int64_t t = 0; t < nt; ++t) { if (h[t*8]) tmin = std::min(tmin, h[t*8]); tmax = std::max(tmax, h[t*8+7]); }
            printf("  kernel span %lld ticks for %lld tiles\n", tmax - tmin, (long long)nt);
            return 0;
        }
        run_onesweep<20, 2>(B, m, shift, false, "onesweep");
        return 0;
    }
    // ---- onesweep variants
#define CHECK() do { } while (0)      /* (the results are covered by tests/test_gpu_parity.py) */
#define CHECKK() do { } while (0)
#define KV(I, W, T, LM, R) run_onesweep<I, W, T, false, LM, R, kKeys>(B, m, shift, false, "keys"); CHECKK();
#define PV(I, W, T, LM, R) run_onesweep<I, W, T, false, LM, R, kPairs>(B, m, shift, false, "pairs"); CHECK();
    KV(24, 2, 512, true, 2) KV(24, 2, 512, false, 2) PV(20, 2, 512, false, 2) PV(20, 2, 512, true, 2)
    return 0;
}
