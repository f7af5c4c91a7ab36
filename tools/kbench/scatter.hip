// Scattered-run write microbenchmark: the output pattern of one radix digit pass on uniform digits
// (tile t writes, for every digit d, a run of L words at region d, offset t*L), without any reads
// or ranking.  Tells how much of a pass's time is the write pattern itself.
//   hipcc --offload-arch=gfx950 -O3 -o tools/kbench/scatter tools/kbench/scatter.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// kRemap = G > 0: workgroups are dealt to the 8 XCDs round robin (blockIdx % 8), so inside every group of 8 G
// workgroups XCD x gets tiles x G ... x G + G - 1: G consecutive tiles -- whose runs are neighbours in every region --
// meet in one L2 and can leave it as whole lines
template <typename T, int kItems, int kRegions = 256, int kRemap = 0>
__global__ __launch_bounds__(512) void scatter_runs(T *out, int64_t n, int L, int64_t region)
{
    int64_t tile = blockIdx.x;
    if (kRemap > 0) {
        constexpr int64_t kGroup = 8 * kRemap;
        const int64_t g = tile / kGroup, r = tile % kGroup;
        if ((g + 1) * kGroup <= (int64_t)gridDim.x) tile = g * kGroup + (r & 7) * kRemap + (r >> 3);
    }
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const int q = k * 512 + threadIdx.x;
        const int d = q / L, i = q - d * L;
        const int64_t o = (int64_t)d * region + tile * L + i;
        if (d < kRegions) out[o] = (T)(o ^ (uint64_t)tile);
    }
}

template <typename T>
__global__ __launch_bounds__(512) void stream_write(T *out, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 512) out[i] = (T)i;
}

template <typename T, int kItems, int kRegions = 256, int kRemap = 0>
int run(const char *name, int64_t n)
{
    const int tileN = 512 * kItems;
    const int L = tileN / kRegions;
    const int64_t ntiles = n / tileN;
    const int64_t region = ntiles * L + 37;          // not a power of two
    T *out;
    CK(hipMalloc(&out, (size_t)(kRegions * region + 64) * sizeof(T)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((scatter_runs<T, kItems, kRegions, kRemap>), dim3((unsigned)ntiles), dim3(512), 0, 0, out, n, L, region);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
    }
    printf("%-28s run %5d B  tiles %6lld : %8.1f us  %7.1f GB/s written\n", name, (int)(L * sizeof(T)), (long long)ntiles,
           best * 1e3, ntiles * tileN * sizeof(T) / best / 1e6);
    best = 1e9;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(stream_write<T>, dim3(256 * 8), dim3(512), 0, 0, out, n);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
    }
    printf("%-28s streaming write          : %8.1f us  %7.1f GB/s\n", name, best * 1e3, n * sizeof(T) / best / 1e6);
    CK(hipFree(out));
    return 0;
}

int main()
{
    const int64_t n = 64ll << 20;
    run<uint64_t, 12>("u64 x 6144/tile", n);
    run<uint64_t, 24>("u64 x 12288/tile", n);
    run<uint64_t, 48>("u64 x 24576/tile", n);
    run<uint64_t, 96>("u64 x 49152/tile", n);
    run<uint32_t, 24>("u32 x 12288/tile", n);
    run<uint32_t, 48>("u32 x 24576/tile", n);
    run<uint32_t, 96>("u32 x 49152/tile", n);
    // wider digits: 2048 regions (11-bit digits), 1024 (10-bit), 512 (9-bit)
    run<uint64_t, 24, 2048>("u64 x 12288/tile, 2048 reg", n);
    run<uint64_t, 48, 2048>("u64 x 24576/tile, 2048 reg", n);
    run<uint32_t, 24, 2048>("u32 x 12288/tile, 2048 reg", n);
    run<uint32_t, 48, 2048>("u32 x 24576/tile, 2048 reg", n);
    run<uint64_t, 24, 1024>("u64 x 12288/tile, 1024 reg", n);
    run<uint32_t, 24, 1024>("u32 x 12288/tile, 1024 reg", n);
    run<uint64_t, 24, 512>("u64 x 12288/tile, 512 reg", n);
    run<uint32_t, 24, 512>("u32 x 12288/tile, 512 reg", n);
    // XCD-aware tile order
    run<uint64_t, 24, 256, 8>("u64 12288 256reg remap8", n);
    run<uint32_t, 24, 256, 8>("u32 12288 256reg remap8", n);
    run<uint64_t, 24, 2048, 8>("u64 12288 2048reg remap8", n);
    run<uint32_t, 24, 2048, 8>("u32 12288 2048reg remap8", n);
    run<uint64_t, 24, 2048, 16>("u64 12288 2048reg remap16", n);
    run<uint32_t, 24, 2048, 16>("u32 12288 2048reg remap16", n);
    run<uint64_t, 24, 2048, 32>("u64 12288 2048reg remap32", n);
    run<uint32_t, 24, 2048, 32>("u32 12288 2048reg remap32", n);
    run<uint64_t, 24, 1024, 16>("u64 12288 1024reg remap16", n);
    run<uint32_t, 24, 1024, 16>("u32 12288 1024reg remap16", n);
    return 0;
}
