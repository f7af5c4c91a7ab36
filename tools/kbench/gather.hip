// Random-gather microbenchmark for DESIGN section 8 ("first doubling round keyed from the text"): G gathers of W
// bytes at pseudo-random positions of an array of S bytes, beside a streaming read of the request list -- what a
// doubling round's key2 gather does.  Compares S = 1 GiB (4-byte ranks of a 256 MiB text: one 64-byte sector per gather
// from HBM) with S = 256 MiB (the text itself, 16 bytes per gather: does it live in the 256 MB last-level cache?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/kbench/gather tools/kbench/gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// the request list: idx[i] = a pseudo-random element index (made once, then streamed like a list of suffixes)
__global__ void make_idx(uint32_t *idx, int64_t g, uint64_t elems, uint64_t seed)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g; i += (int64_t)gridDim.x * blockDim.x)
        idx[i] = (uint32_t)(mix((uint64_t)i + seed) % elems);
}

template <int kWords>           // kWords dwords per gather, at a 4-byte aligned position
__global__ __launch_bounds__(1024) void gather(const uint32_t *__restrict__ a, const uint32_t *__restrict__ idx, int64_t g,
                                               uint32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i >= g) return;
    const uint32_t p = idx[i];
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kWords; ++k) acc += a[(size_t)p + k];
    out[i] = acc;
}

template <int kWords>
int run(const char *name, size_t bytes, int64_t g)
{
    uint32_t *a, *idx, *out;
    CK(hipMalloc(&a, bytes + 64));
    CK(hipMalloc(&idx, (size_t)g * 4));
    CK(hipMalloc(&out, (size_t)g * 4));
    CK(hipMemset(a, 1, bytes + 64));
    hipLaunchKernelGGL(make_idx, dim3(4096), dim3(256), 0, 0, idx, g, (uint64_t)(bytes / 4 - kWords), 12345ull);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather<kWords>, dim3((unsigned)((g + 1023) / 1024)), dim3(1024), 0, 0, a, idx, g, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s %6.1f M gathers of %2d B from %5zu MiB: %7.3f ms = %6.1f G gathers/s (%5.2f TB/s of 64-byte sectors)\n", name,
           g / 1e6, 4 * kWords, bytes >> 20, best, g / best / 1e6, g * 64.0 / best / 1e9);
    CK(hipFree(a)); CK(hipFree(idx)); CK(hipFree(out));
    return 0;
}

int main()
{
    const int64_t g = 130 << 20;
    if (run<1>("ranks of a 256 MiB text (ISA gather)", (size_t)1 << 30, g)) return 1;
    if (run<4>("16 text bytes of a 256 MiB text", (size_t)256 << 20, g)) return 1;
    if (run<1>("4 bytes of a 256 MiB array", (size_t)256 << 20, g)) return 1;
    if (run<1>("4 bytes of a 64 MiB array", (size_t)64 << 20, g)) return 1;
    if (run<4>("16 bytes of a 1 GiB array", (size_t)1 << 30, g)) return 1;
    return 0;
}
