// stream.hip -- streaming-kernel micro-benchmarks (developer tool): what does it cost to read
// 64 Mi u64 keys and detect adjacent-equal pairs, in increasing levels of work?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int THREADS, int K, int LEVEL>
__global__ __launch_bounds__(THREADS) void k_striped(const uint64_t *__restrict__ keys, int64_t m, unsigned long long *out)
{
    const int lane = lane_id(), w = threadIdx.x >> 6;
    const int64_t wb = ((int64_t)blockIdx.x * (THREADS / 64) + w) * (64 * K);
    const uint64_t *kp = keys + wb;
    uint64_t ck[K];
#pragma unroll
    for (int k = 0; k < K; ++k) ck[k] = (wb + k * 64 + lane < m) ? kp[k * 64 + lane] : 0;
    unsigned long long acc = 0;
    if (LEVEL == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= ck[k];
    } else {
        uint64_t prevlast = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)ck[k], 0x138, 0xf, 0xf, false);
            const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(ck[k] >> 32), 0x138, 0xf, 0xf, false);
            uint64_t pk = ((uint64_t)phi << 32) | plo;
            if (lane == 0) pk = prevlast;
            const uint64_t H = __ballot(ck[k] != pk);
            prevlast = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ck[k] >> 32), 63) << 32) |
                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ck[k], 63);
            if (LEVEL == 1) acc += __popcll(H);
            if (LEVEL >= 2) {   // per-lane: last head at or below me
                const uint64_t hm = H & ((2ull << lane) - 1);
                acc += hm ? 63 - __builtin_clzll(hm) : 0;
            }
        }
    }
    if (LEVEL <= 1) { if (lane == 0 && acc == (unsigned long long)m) out[0] = acc; }
    else { if (acc == (unsigned long long)m) out[0] = acc; }
}

// LEVEL 3: + ticket + syncs; 4: + halo scalar loads & variable shift; 5: + masks parked in LDS and re-read in a second phase
template <int THREADS, int K, int LEVEL>
__global__ __launch_bounds__(THREADS) void k_seglike(const uint64_t *__restrict__ keys, int64_t m, unsigned long long *out, unsigned *ticket, int kshift)
{
    __shared__ unsigned s_tile;
    __shared__ uint64_t mH[THREADS / 64][K];
    __shared__ long long wagg[THREADS / 64];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t wb = ((int64_t)s_tile * (THREADS / 64) + w) * (64 * K);
    const uint64_t *kp = keys + wb;
    uint64_t ck[K];
#pragma unroll
    for (int k = 0; k < K; ++k) ck[k] = (wb + k * 64 + lane < m) ? kp[k * 64 + lane] >> (LEVEL >= 4 ? kshift : 0) : 0;
    uint64_t prevlast = 0;
    if (LEVEL >= 4) prevlast = wb > 0 ? keys[wb - 1] >> kshift : 0;
    long long agg = 0;
    unsigned long long acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)ck[k], 0x138, 0xf, 0xf, false);
        const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(ck[k] >> 32), 0x138, 0xf, 0xf, false);
        uint64_t pk = ((uint64_t)phi << 32) | plo;
        if (lane == 0) pk = prevlast;
        const uint64_t H = __ballot(ck[k] != pk);
        prevlast = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ck[k] >> 32), 63) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ck[k], 63);
        agg += __popcll(H);
        if (LEVEL >= 5) { if (lane == 0) mH[w][k] = H; }
        else { const uint64_t hm = H & ((2ull << lane) - 1); acc += hm ? 63 - __builtin_clzll(hm) : 0; }
    }
    if (lane == 0) wagg[w] = agg;
    __syncthreads();
    long long pre = 0;
    for (int i = 0; i < w; ++i) pre += wagg[i];
    if (LEVEL >= 5) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t H = mH[w][k];
            if (H) { const uint64_t hm = H & ((2ull << lane) - 1); acc += hm ? 63 - __builtin_clzll(hm) : 0; }
        }
    }
    if (acc + pre == (unsigned long long)m) out[0] = acc;
}

template <int THREADS, int PAIRS>
__global__ __launch_bounds__(THREADS) void k_vec16(const uint64_t *__restrict__ keys, int64_t m, unsigned long long *out)
{
    const ulonglong2 *k2 = reinterpret_cast<const ulonglong2 *>(keys);
    const int64_t base = (int64_t)blockIdx.x * THREADS * PAIRS;
    unsigned long long acc = 0;
    ulonglong2 v[PAIRS];
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) { const int64_t i = base + j * THREADS + threadIdx.x; v[j] = (2 * i + 1 < m) ? k2[i] : ulonglong2{0, 0}; }
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) acc += (v[j].x == v[j].y);
    if (acc == (unsigned long long)m) out[0] = acc;
}

template <class F> float time_it(F &&f, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float sum = 0;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms; }
    return sum / reps;
}
#define RUN(NAME, KERN, THREADS, PERBLOCK) { const int64_t g = (m + (PERBLOCK) - 1) / (PERBLOCK); float ms = time_it([&]() { hipLaunchKernelGGL(KERN, dim3((unsigned)g), dim3(THREADS), 0, 0, keys, m, out); }); printf("%-44s %8.1f us  %7.1f GB/s\n", NAME, ms * 1e3, m * 8.0 / ms / 1e6); }
int main()
{
    const int64_t m = 64ll << 20;
    uint64_t *keys; unsigned long long *out;
    CK(hipMalloc(&keys, m * 8)); CK(hipMalloc(&out, 64)); CK(hipMemset(keys, 1, m * 8));
    RUN("vec16 256thr x4 (pure 16B loads)", (k_vec16<256, 4>), 256, 256 * 4 * 2)
    RUN("vec16 256thr x8", (k_vec16<256, 8>), 256, 256 * 8 * 2)
    RUN("striped 256thr K=16 L0 (xor)", (k_striped<256, 16, 0>), 256, 256 * 16)
    RUN("striped 1024thr K=16 L0", (k_striped<1024, 16, 0>), 1024, 1024 * 16)
    RUN("striped 256thr K=8 L0", (k_striped<256, 8, 0>), 256, 256 * 8)
    RUN("striped 256thr K=16 L1 (dpp+ballot+popc)", (k_striped<256, 16, 1>), 256, 256 * 16)
    RUN("striped 1024thr K=16 L1", (k_striped<1024, 16, 1>), 1024, 1024 * 16)
    RUN("striped 256thr K=16 L2 (+per-lane msb)", (k_striped<256, 16, 2>), 256, 256 * 16)
    RUN("striped 1024thr K=16 L2", (k_striped<1024, 16, 2>), 1024, 1024 * 16)
    RUN("striped 512thr K=8 L2", (k_striped<512, 8, 2>), 512, 512 * 8)
    unsigned *ticket; CK(hipMalloc(&ticket, 64));
#define RUNS(NAME, KERN, THREADS, PERBLOCK) { const int64_t g = (m + (PERBLOCK) - 1) / (PERBLOCK); float ms = time_it([&]() { CK(hipMemsetAsync(ticket, 0, 4)); hipLaunchKernelGGL(KERN, dim3((unsigned)g), dim3(THREADS), 0, 0, keys, m, out, ticket, 26); }); printf("%-44s %8.1f us  %7.1f GB/s\n", NAME, ms * 1e3, m * 8.0 / ms / 1e6); }
    RUNS("seglike 1024thr K=16 L3 (ticket+sync)", (k_seglike<1024, 16, 3>), 1024, 1024 * 16)
    RUNS("seglike 1024thr K=16 L4 (+halo,+shift)", (k_seglike<1024, 16, 4>), 1024, 1024 * 16)
    RUNS("seglike 1024thr K=16 L5 (+LDS masks)", (k_seglike<1024, 16, 5>), 1024, 1024 * 16)
    RUNS("seglike 256thr K=16 L3", (k_seglike<256, 16, 3>), 256, 256 * 16)
    RUNS("seglike 256thr K=16 L5", (k_seglike<256, 16, 5>), 256, 256 * 16)
    RUNS("seglike 512thr K=16 L5", (k_seglike<512, 16, 5>), 512, 512 * 16)
    return 0;
}
