// Random-access microbenchmarks on MI355X: what bounds the k-mer table's insert kernel?
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_random.hip -o gpurun_out/microbench_random
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint64_t u64; typedef uint32_t u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ u64 mix64(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
struct Slot { u64 key; u32 count; u32 pad; };

// mode 0: load key only; 1: atomicAdd only; 2: load + atomicAdd (insert pattern); 3: load + plain store (non-atomic RMW)
// 4: load + atomicAdd on 64-bit word; idx = mulhi(mix(i*stride+seed) , cap) restricted to a window of `window` slots that
// moves with i (window == cap -> fully random)
template <int MODE>
__global__ __launch_bounds__(256) void rand_kernel(Slot* t, u64 cap, u64 n, u64 window, u64 seed, u64* sink) {
    u64 acc = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        u64 h = mix64(i + seed);
        u64 base = window >= cap ? 0 : (u64)((unsigned __int128)i * (cap - window) / n);
        u64 s = base + (u64)(((unsigned __int128)h * window) >> 64);
        if (MODE == 0) acc += t[s].key;
        if (MODE == 1) atomicAdd(&t[s].count, 1u);
        if (MODE == 2) { acc += t[s].key; atomicAdd(&t[s].count, 1u); }
        if (MODE == 3) { u32 c = t[s].count; t[s].count = c + 1; }
        if (MODE == 4) { acc += t[s].key; atomicAdd((unsigned long long*)&t[s].key, 1ull); }
    }
    if (acc == 0x1234567) *sink = acc;
}
template <int MODE> float run(Slot* t, u64 cap, u64 n, u64 window, u64* sink) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(rand_kernel<MODE>, dim3(8192), dim3(256), 0, 0, t, cap, n / 8, window, 1, sink);   // warm
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(rand_kernel<MODE>, dim3(8192), dim3(256), 0, 0, t, cap, n, window, 7, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms;
}
int main() {
    const u64 n = 1ull << 29;   // 5.4e8 accesses
    u64* sink; CHECK(hipMalloc(&sink, 8));
    const char* names[5] = {"load8", "atomicAdd32", "load+atomicAdd32", "load+store (non-atomic)", "load+atomicAdd64"};
    for (u64 mb : {32ull, 128ull, 512ull, 4096ull, 32768ull}) {
        u64 cap = mb * 1024 * 1024 / sizeof(Slot);
        Slot* t; CHECK(hipMalloc(&t, cap * sizeof(Slot))); CHECK(hipMemset(t, 0, cap * sizeof(Slot)));
        for (u64 window : {cap, cap / 256 ? cap / 256 : 1, (u64)16384}) {
            if (window > cap) continue;
            float ms[5] = {run<0>(t, cap, n, window, sink), run<1>(t, cap, n, window, sink), run<2>(t, cap, n, window, sink),
                           run<3>(t, cap, n, window, sink), run<4>(t, cap, n, window, sink)};
            printf("table %6llu MiB window %10llu slots (%8.2f MiB):", (unsigned long long)mb, (unsigned long long)window, window * 16.0 / 1048576);
            for (int m = 0; m < 5; ++m) printf("  %s %.2e/s", names[m], n / (ms[m] * 1e-3));
            printf("\n"); fflush(stdout);
        }
        CHECK(hipFree(t));
    }
    return 0;
}
