// The ceiling of the partition pass (radix.hip radix_scatter_kernel), with the ranking work taken out: what rate does the card reach
// for the pass's MEMORY PATTERN alone?  A workgroup takes a tile of 4096 records (8-byte key + 4-byte value, as the edge sort's pairs),
// reads it with coalesced loads and writes it as 256 stretches of 16 records -- 128 bytes of keys, 64 bytes of values each -- one
// stretch into each of 256 output streams that lie n / 256 records apart: exactly what the pass does when every digit holds 16 of a
// tile's records, but the destination of a record is i -> (i % 256 streams, tile * 16 + i / 256): no histogram, no ballots, no LDS
// reorder.  Variants: tile order plain or XCD-aware (workgroup i takes tile (i % 8) * (tiles / 8) + i / 8, as the pass does);
// input loads plain or non-temporal; and, for scale, a plain copy of the same bytes.
//   build: hipcc -O3 --offload-arch=gfx950 tools/microbench_scatter.hip -o tools/microbench_scatter     run: tools/microbench_scatter [n = 1611246430]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint64_t u64; typedef uint32_t u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr u32 TILE = 4096, STREAMS = 256, STRETCH = TILE / STREAMS, BLOCK = 256;

template <bool XCD, bool NT>
__global__ __launch_bounds__(BLOCK) void scatter_kernel(const u64* __restrict__ kin, const u32* __restrict__ vin, u64 n_tiles, u64 stream_len,
                                                        u64* __restrict__ kout, u32* __restrict__ vout, u32 per_xcd) {
    const u64 tile = XCD ? (u64)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
    if (tile >= n_tiles) return;
    const u64 base = tile * TILE;
    u64 k[16]; u32 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {            // row j: BLOCK consecutive records
        const u64 i = base + (u64)j * BLOCK + threadIdx.x;
        k[j] = NT ? __builtin_nontemporal_load(kin + i) : kin[i];
        v[j] = NT ? __builtin_nontemporal_load(vin + i) : vin[i];
    }
    // record (row j, thread t) -> stream s = j * 16 + t / 16, place t % 16 inside the tile's stretch: 16 consecutive threads write one
    // 128-byte stretch of keys (and 64 bytes of values)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const u32 s = (u32)j * 16 + threadIdx.x / STRETCH, p = threadIdx.x % STRETCH;
        const u64 at = (u64)s * stream_len + tile * STRETCH + p;
        kout[at] = k[j]; vout[at] = v[j];
    }
}
__global__ __launch_bounds__(BLOCK) void copy_kernel(const u64* __restrict__ kin, const u32* __restrict__ vin, u64 n, u64* __restrict__ kout, u32* __restrict__ vout) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) { kout[i] = kin[i]; vout[i] = vin[i]; }
}

int main(int argc, char** argv) {
    u64 n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1611246430ull;
    const u64 n_tiles = n / TILE;
    n = n_tiles * TILE;
    const u64 stream_len = n_tiles * STRETCH;
    u64 *kin, *kout; u32 *vin, *vout;
    CHECK(hipMalloc(&kin, n * 8)); CHECK(hipMalloc(&kout, n * 8)); CHECK(hipMalloc(&vin, n * 4)); CHECK(hipMalloc(&vout, n * 4));
    CHECK(hipMemset(kin, 1, n * 8)); CHECK(hipMemset(vin, 2, n * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const double bytes = (double)n * 24.0;
    const u32 per_xcd = (u32)((n_tiles + 7) / 8);
    auto time = [&](const char* what, auto launch) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
        }
        CHECK(hipGetLastError());
        printf("%-78s %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", what, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
    };
    printf("%llu pairs of 8 + 4 bytes; read + write = %.2f GB per pass\n", (unsigned long long)n, bytes / 1e9);
    time("copy (grid-stride, the same bytes)", [&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 32), dim3(BLOCK), 0, 0, kin, vin, n, kout, vout); });
    time("scatter: 4096-record tiles -> 256 streams x 16 records, tiles in plain order", [&] { hipLaunchKernelGGL((scatter_kernel<false, false>), dim3((u32)n_tiles), dim3(BLOCK), 0, 0, kin, vin, n_tiles, stream_len, kout, vout, per_xcd); });
    time("scatter: the same, XCD-aware tile order", [&] { hipLaunchKernelGGL((scatter_kernel<true, false>), dim3(per_xcd * 8), dim3(BLOCK), 0, 0, kin, vin, n_tiles, stream_len, kout, vout, per_xcd); });
    time("scatter: XCD-aware tile order, non-temporal input loads (what the pass does)", [&] { hipLaunchKernelGGL((scatter_kernel<true, true>), dim3(per_xcd * 8), dim3(BLOCK), 0, 0, kin, vin, n_tiles, stream_len, kout, vout, per_xcd); });
    time("scatter: plain order, non-temporal input loads", [&] { hipLaunchKernelGGL((scatter_kernel<false, true>), dim3((u32)n_tiles), dim3(BLOCK), 0, 0, kin, vin, n_tiles, stream_len, kout, vout, per_xcd); });
    return 0;
}
