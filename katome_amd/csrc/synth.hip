// synth.hip -- deterministic synthetic read generator, on device (bench + tests only).
// Definition in DESIGN.md "Synthetic workload"; the test suite holds an independent CPU statement of
// the same definition and compares the two byte for byte.
// Output is what the host ingest (reference builder.rs:142-165) would hand to the build:
// reads 2-bit packed (compress_node bit order, compress.rs:55-73) and one skip flag per read
// for reads that hold a non-ACGT byte (builder.rs:155-157).
#include "common.h"

namespace katome {

struct SynthParams {
    u64 BG, BR, BE, BN, first_read, n_reads, genome_len, thr;
    u32 read_len, stride, n_pct;
};

__global__ __launch_bounds__(BLOCK) void synth_kernel(SynthParams p, uint8_t* __restrict__ packed, uint8_t* __restrict__ skip) {
    const u64 total = p.n_reads * p.stride;
    const u64 L = p.read_len;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (u64)gridDim.x * BLOCK) {
        const u64 rl = i / p.stride;
        const u32 jb = (u32)(i - rl * p.stride);
        const u64 r = p.first_read + rl;
        const u64 start = splitmix64(p.BR + 2 * r) % (p.genome_len - L + 1);
        const u64 strand = splitmix64(p.BR + 2 * r + 1) & 1;
        u32 byte = 0;
#pragma unroll
        for (u32 q = 0; q < 4; ++q) {
            const u64 j = (u64)jb * 4 + q;
            u64 b = 0;
            if (j < L) {
                b = strand ? 3 - (splitmix64(p.BG + start + (L - 1 - j)) & 3) : (splitmix64(p.BG + start + j) & 3);
                const u64 x = splitmix64(p.BE + r * L + j);
                if ((x >> 40) < p.thr) b = (b + 1 + ((x & 0xFFFF) % 3)) & 3;
            }
            byte = (byte << 2) | (u32)b;
        }
        packed[i] = (uint8_t)byte;
        if (jb == 0 && skip) {
            const u64 u = splitmix64(p.BN + r);
            skip[rl] = (p.n_pct && (u % 100) < p.n_pct) ? 1 : 0;
        }
    }
}

int launch_synth(uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len, double err_rate,
                 uint32_t n_inject_percent, uint8_t* d_packed, uint8_t* d_skip, hipStream_t stream) {
    if (n_reads == 0) return KATOME_OK;
    if (read_len == 0 || genome_len < read_len) { set_error("synth: genome shorter than a read"); return KATOME_E_ARG; }
    SynthParams p;
    p.BG = splitmix64(0x6B61746F6D650001ull); p.BR = splitmix64(0x6B61746F6D650002ull);
    p.BE = splitmix64(0x6B61746F6D650003ull); p.BN = splitmix64(0x6B61746F6D650004ull);
    p.first_read = first_read; p.n_reads = n_reads; p.genome_len = genome_len;
    p.thr = (u64)(err_rate * 16777216.0);
    p.read_len = read_len; p.stride = (read_len + 3) / 4; p.n_pct = n_inject_percent;
    hipLaunchKernelGGL(synth_kernel, dim3(grid_for(n_reads * p.stride, BLOCK)), dim3(BLOCK), 0, stream, p, d_packed, d_skip);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

}  // namespace katome
