// standardize.hip -- Standardizable for PtGraph (reference src/katome/algorithms/standardizer.rs:41-128): the two weight
// passes that assemble_with_graph runs between its prunings (asm/basic_assembler.rs:63-70).
//
// standardize_contigs (72-122): every contig -- an out-edge of an AMBIGUOUS vertex (in > 1 or out > 1, or in = 0 with
// out >= 1: pt_graph.rs:54-62) followed through first_edge(Outgoing) while the vertex reached has exactly one out-edge
// and is not ambiguous -- gets the rounded mean of its weights on all its edges.  Contigs never share an edge and nothing
// is re-numbered, so one thread per contig reproduces the sequential loop exactly (same f64 division and round()).
// standardize_edges (42-70): weights scaled by (genome length - k) / (sum of weights - sum of those under the threshold),
// rounded, lifted to 1 where a weight at or above the threshold would vanish, then remove_weak_edges(1).
#include <math.h>

#include "common.h"

namespace katome {
namespace {

typedef uint32_t u32;

// node_deg[v]: in-degree (low half) / out-degree (high half); out_edge[v]: an out-edge of v (THE out-edge where there is one)
__global__ __launch_bounds__(BLOCK) void adjacency_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, u64 E,
                                                          u64* __restrict__ node_deg, u32* __restrict__ out_edge) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        const u64 a = src[e], b = dst[e];
        atomicAdd((unsigned long long*)&node_deg[a], 1ull << 32);
        atomicAdd((unsigned long long*)&node_deg[b], 1ull);
        out_edge[a] = (u32)e;
    }
}
__device__ __forceinline__ bool ambiguous(u64 deg) {
    const u32 in = (u32)deg, out = (u32)(deg >> 32);
    return in > 1 || out > 1 || (in == 0 && out >= 1);
}
// One word per vertex so that a step of a contig walk is one look-up: a vertex a contig runs through (exactly one out-edge,
// not ambiguous) holds (the vertex that edge leads to) << 32 | that edge; an ambiguous vertex AMBIGUOUS; any other END.
constexpr u64 AMBIGUOUS = ~0ull, END = ~0ull - 1;
__global__ __launch_bounds__(BLOCK) void node_word_kernel(u64 N, const u64* __restrict__ node_deg, const u32* __restrict__ out_edge,
                                                          const u64* __restrict__ dst, u64* __restrict__ word) {
    for (u64 v = (u64)blockIdx.x * BLOCK + threadIdx.x; v < N; v += (u64)gridDim.x * BLOCK) {
        const u64 deg = node_deg[v];
        u64 w = END;
        if (ambiguous(deg)) w = AMBIGUOUS;
        else if ((u32)(deg >> 32) == 1) { const u32 oe = out_edge[v]; w = (dst[oe] << 32) | oe; }
        word[v] = w;
    }
}
__global__ __launch_bounds__(BLOCK) void contig_mean_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, u64 E,
                                                            const u64* __restrict__ word, u32* __restrict__ weight) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        if (word[src[e]] != AMBIGUOUS) continue;              // contigs start at the out-edges of ambiguous vertices
        u64 sum = weight[e], len = 1;
        const u64 first = word[dst[e]];
        for (u64 w = first; w < END; w = word[w >> 32]) { sum += weight[(u32)w]; ++len; }
        const u32 mean = (u32)round((double)sum / (double)len);  // (sum as f64 / contig.len() as f64).round() as EdgeWeight
        weight[e] = mean;
        u64 w = first;
        for (u64 i = 1; i < len; ++i) { weight[(u32)w] = mean; w = word[w >> 32]; }
    }
}
__global__ __launch_bounds__(BLOCK) void weight_sums_kernel(const u32* __restrict__ weight, u64 E, u32 threshold, u64* __restrict__ sums) {
    __shared__ u64 s_all, s_low;
    if (threadIdx.x == 0) { s_all = 0; s_low = 0; }
    __syncthreads();
    u64 a = 0, l = 0;
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        const u32 w = weight[e];
        a += w;
        if (w < threshold) l += w;
    }
    if (a) atomicAdd((unsigned long long*)&s_all, (unsigned long long)a);
    if (l) atomicAdd((unsigned long long*)&s_low, (unsigned long long)l);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_all) atomicAdd((unsigned long long*)&sums[0], (unsigned long long)s_all);
        if (s_low) atomicAdd((unsigned long long*)&sums[1], (unsigned long long)s_low);
    }
}
__global__ __launch_bounds__(BLOCK) void scale_weights_kernel(u32* __restrict__ weight, u64 E, double p, u32 threshold) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        const u32 w = weight[e];
        const double scaled = round((double)w * p);
        const u32 nw = scaled >= 4294967295.0 ? 4294967295u : scaled > 0 ? (u32)scaled : 0u;   // `as EdgeWeight` saturates, NaN -> 0
        weight[e] = (nw == 0 && w >= threshold) ? 1u : nw;
    }
}

}  // namespace

int dev_standardize_contigs(const uint64_t* src, const uint64_t* dst, uint32_t* weight, uint64_t E, uint64_t N, hipStream_t stream) {
    if (E == 0) return KATOME_OK;
    if (E >= 0xFFFFFFFFull) { set_error("standardize_contigs: more than 2^32 edges on one GPU"); return KATOME_E_UNSUPPORTED; }
    DevBuf node_deg(stream), out_edge(stream);
    KCHECK(node_deg.alloc((N + 1) * 8)); KCHECK(out_edge.alloc((N + 1) * 4));
    KCHECK_HIP(hipMemsetAsync(node_deg.p, 0, N * 8, stream));
    const dim3 grid(grid_for(E, BLOCK, 256u * 32u)), blk(BLOCK);
    hipLaunchKernelGGL(adjacency_kernel, grid, blk, 0, stream, src, dst, E, node_deg.as<u64>(), out_edge.as<u32>());
    DevBuf word(stream);
    KCHECK(word.alloc((N + 1) * 8));
    hipLaunchKernelGGL(node_word_kernel, dim3(grid_for(N, BLOCK, 256u * 32u)), blk, 0, stream, N, node_deg.as<u64>(), out_edge.as<u32>(), dst,
                       word.as<u64>());
    hipLaunchKernelGGL(contig_mean_kernel, grid, blk, 0, stream, src, dst, E, word.as<u64>(), weight);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

// the scaling half of standardize_edges; the caller follows with remove_weak_edges(1)
int dev_standardize_scale(uint32_t* weight, uint64_t E, uint64_t original_genome_length, uint32_t k, uint32_t threshold, hipStream_t stream) {
    if (original_genome_length < k) { set_error("standardize_edges: original_genome_length < k"); return KATOME_E_ARG; }
    if (E == 0) return KATOME_OK;
    DevBuf sums(stream);
    KCHECK(sums.alloc(16));
    KCHECK_HIP(hipMemsetAsync(sums.p, 0, 16, stream));
    const dim3 grid(grid_for(E, BLOCK, 256u * 16u)), blk(BLOCK);
    hipLaunchKernelGGL(weight_sums_kernel, grid, blk, 0, stream, weight, E, threshold, sums.as<u64>());
    u64 h[2] = {0, 0};
    KCHECK_HIP(hipMemcpyAsync(h, sums.p, 16, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    const double p = (double)(original_genome_length - k) / (double)(h[0] - h[1]);      // calculate_standardization_ratio (124-128)
    hipLaunchKernelGGL(scale_weights_kernel, grid, blk, 0, stream, weight, E, p, threshold);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

}  // namespace katome
