// What an LDS operation on random slots costs on gfx950, per wave64 instruction, when a workgroup of 1024 owns its CU (the shape of the
// LDS counting kernels of table.hip): plain 8-byte reads, 64-bit compare-and-swap (returning), 32- and 64-bit adds (returning or not), on
// 13312 / 19456 slots.  Prints cycles of CU time per wave instruction (= kernel time x clock / (wave instructions per CU)).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench_lds_atomics.hip -o tools/microbench_lds_atomics && tools/microbench_lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;
constexpr u32 THREADS = 1024, ITERS = 4096;
__device__ __forceinline__ u32 next(u32& x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int OP, u32 SLOTS>
__global__ __launch_bounds__(THREADS) void bench(u64* out) {
    extern __shared__ u64 lds[];
    u32* lds32 = reinterpret_cast<u32*>(lds);
    for (u32 i = threadIdx.x; i < SLOTS; i += THREADS) lds[i] = 0;
    __syncthreads();
    u32 x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    u64 acc = 0;
    for (u32 it = 0; it < ITERS; ++it) {
        const u32 s = (u32)(((u64)next(x) * SLOTS) >> 32);
        if (OP == 0) acc += lds[s];                                                    // plain 8-byte read
        if (OP == 1) acc += atomicCAS(&lds[s], 0ull, (u64)x | 1ull);                    // 64-bit CAS, returning
        if (OP == 2) atomicAdd(&lds32[s], 1u);                                          // 32-bit add, not returning
        if (OP == 3) acc += atomicAdd(&lds32[s], 1u);                                   // 32-bit add, returning
        if (OP == 4) atomicAdd(&lds[s], 1ull);                                          // 64-bit add, not returning
        if (OP == 5) acc += atomicAdd(&lds[s], 1ull);                                   // 64-bit add, returning
        if (OP == 6) { const u64 c = atomicCAS(&lds[s], 0ull, (u64)x | 1ull); acc += c; atomicAdd(&lds32[2 * SLOTS + s], 1u); }   // CAS then 32-bit add (lds_count_kernel's pair)
        if (OP == 7) { const u64 c = lds[s]; acc += c; if (c == 0) acc += atomicCAS(&lds[s], 0ull, (u64)x | 1ull); else atomicAdd(&lds[s], 1ull); }   // read, then CAS or add (packed kernel)
        if (OP == 8) acc += x;                                                          // nothing: the loop's own cost
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

template <int OP, u32 SLOTS> void run(const char* what, u64* d_out, double clock_hz) {
    const size_t bytes = (size_t)SLOTS * (OP == 6 ? 12 : 8);
    hipFuncSetAttribute((const void*)bench<OP, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    bench<OP, SLOTS><<<256, THREADS, bytes>>>(d_out);
    hipEventRecord(a);
    bench<OP, SLOTS><<<256, THREADS, bytes>>>(d_out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double wave_instr_per_cu = (double)(THREADS / 64) * ITERS;
    printf("%-62s %6u slots  %8.3f ms  %7.1f cycles of CU time per wave instruction (%.2f per lane)\n", what, SLOTS, ms,
           ms * 1e-3 * clock_hz / wave_instr_per_cu, ms * 1e-3 * clock_hz / wave_instr_per_cu / 64);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clock_hz = p.clockRate * 1e3;
    printf("%s, %d CUs, %.0f MHz; one workgroup of %u per CU, %u operations per thread\n", p.name, p.multiProcessorCount, clock_hz / 1e6, THREADS, ITERS);
    u64* d; hipMalloc(&d, 64);
    run<8, 13312>("loop only (xorshift + mulhi)", d, clock_hz);
    run<0, 13312>("plain 8-byte read", d, clock_hz);
    run<0, 19456>("plain 8-byte read", d, clock_hz);
    run<1, 13312>("64-bit compare-and-swap, returning", d, clock_hz);
    run<1, 19456>("64-bit compare-and-swap, returning", d, clock_hz);
    run<2, 13312>("32-bit add", d, clock_hz);
    run<3, 13312>("32-bit add, returning", d, clock_hz);
    run<4, 13312>("64-bit add", d, clock_hz);
    run<5, 13312>("64-bit add, returning", d, clock_hz);
    run<6, 13312>("64-bit CAS + 32-bit add (12-byte slots)", d, clock_hz);
    run<7, 19456>("8-byte read, then CAS or 64-bit add (8-byte slots)", d, clock_hz);
    hipFree(d);
    return 0;
}
