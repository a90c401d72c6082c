// shrink.hip -- Shrinkable::shrink for PtGraph (reference src/katome/algorithms/shrinker.rs:165-209, labels merged as
// EdgeSlice::merge does, slices.rs:23-34): every maximal straight path -- consecutive edges whose inner vertices have
// exactly one incoming and one outgoing edge -- becomes one edge that spells the whole path (compress_edge format,
// compress.rs:250-271) and carries the weight of the path's FIRST edge (shrinker.rs:181,200); the inner vertices, left
// without edges, are removed (remove_single_vertices, shrinker.rs:172).
//
// The reference reaches that result by a sequential depth-first traversal whose order only decides the numbering of
// what comes out -- and, for the parts of a graph that no vertex without incoming edges can reach, where a path gets
// cut ("shrink").  Here the paths are found from the degrees alone:
//   * a vertex is INNER when in-degree = out-degree = 1 and its edges are not one self-loop;
//   * an edge whose source is not inner is the HEAD of a path: one thread follows it to the first non-inner vertex;
//   * edges no head reaches lie on cycles made of inner vertices only: such a cycle becomes a self-loop at its vertex
//     with the smallest id (what the reference's traversal does when it enters the cycle there).
// Output order (ours): merged edges in the order of their head edges, surviving vertices in their old order.
#include <algorithm>
#include <chrono>
#include <new>

#include "common.h"
#include "shrink_exact.h"

namespace katome {
namespace {

typedef uint32_t u32;
constexpr u32 NO_EDGE = 0xFFFFFFFFu;
constexpr int ITEMS = 8;

// degree words (in-degree low half, out-degree high half) and, per node, one incoming and one outgoing edge (the only
// one where the degree is 1, which is all the walks ask for)
__global__ __launch_bounds__(BLOCK) void adjacency_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, u64 E,
                                                          u64* __restrict__ node_deg, u32* __restrict__ in_edge, u32* __restrict__ out_edge) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        const u64 a = src[e], b = dst[e];
        atomicAdd((unsigned long long*)&node_deg[a], 1ull << 32);
        atomicAdd((unsigned long long*)&node_deg[b], 1ull);
        out_edge[a] = (u32)e;
        in_edge[b] = (u32)e;
    }
}
__device__ __forceinline__ bool is_inner(u64 v, const u64* node_deg, const u32* in_edge, const u32* out_edge) {
    return node_deg[v] == ((1ull << 32) | 1ull) && in_edge[v] != out_edge[v];
}

// One word per node so that a step of a walk is one look-up: for an inner vertex (the vertex its out-edge leads to) << 32 |
// that out-edge, NOT_INNER for every other vertex.
constexpr u64 NOT_INNER = ~0ull;
__global__ __launch_bounds__(BLOCK) void node_word_kernel(u64 N, const u64* __restrict__ node_deg, const u32* __restrict__ in_edge,
                                                          const u32* __restrict__ out_edge, const u64* __restrict__ dst, u64* __restrict__ word) {
    for (u64 v = (u64)blockIdx.x * BLOCK + threadIdx.x; v < N; v += (u64)gridDim.x * BLOCK) {
        u64 w = NOT_INNER;
        if (is_inner(v, node_deg, in_edge, out_edge)) { const u32 oe = out_edge[v]; w = (dst[oe] << 32) | oe; }
        word[v] = w;
    }
}

// flag[e] = 1 for head edges; every edge a head's walk passes gets covered[e] = 1
__global__ __launch_bounds__(BLOCK) void head_walk_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, u64 E,
                                                          const u64* __restrict__ word, u32* __restrict__ flag, u32* __restrict__ covered) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        const bool head = word[src[e]] == NOT_INNER;
        flag[e] = head ? 1u : 0u;
        if (!head) continue;
        covered[e] = 1;
        for (u64 w = word[dst[e]]; w != NOT_INNER; w = word[w >> 32]) covered[(u32)w] = 1;
    }
}
// cycles of inner vertices: the edge leaving the cycle's smallest vertex becomes its head
__global__ __launch_bounds__(BLOCK) void cycle_head_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, u64 E,
                                                           const u64* __restrict__ word, const u32* __restrict__ covered, u32* __restrict__ flag) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        if (covered[e]) continue;
        const u64 start = src[e];
        bool smallest = true;
        for (u64 cur = dst[e]; cur != start; cur = word[cur] >> 32) if (cur < start) { smallest = false; break; }   // (all inner)
        if (smallest) flag[e] = 1;
    }
}

// flagged positions, ascending: count / scan / write (block sums, one scan workgroup, ordered write)
__global__ __launch_bounds__(BLOCK) void flag_count_kernel(const u32* __restrict__ flag, u64 n, u32* __restrict__ counts) {
    __shared__ u32 total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const u64 base = ((u64)blockIdx.x * BLOCK + threadIdx.x) * ITEMS;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) c += (base + j < n && flag[base + j] != 0);
    if (c) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
// out_pos[rank] = position (when given); rank_of[position] = rank among the flagged (when given; others keep NO_EDGE)
__global__ __launch_bounds__(BLOCK) void flag_write_kernel(const u32* __restrict__ flag, u64 n, const u64* __restrict__ block_offs,
                                                           u32* __restrict__ out_pos, u32* __restrict__ rank_of) {
    __shared__ u32 wsum[BLOCK / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = ((u64)blockIdx.x * BLOCK + tid) * ITEMS;
    bool f[ITEMS]; u32 c = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { f[j] = base + j < n && flag[base + j] != 0; c += f[j]; }
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < wave; ++w) woff += wsum[w];
    u64 pos = block_offs[blockIdx.x] + woff + incl - c;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (f[j]) { if (out_pos) out_pos[pos] = (u32)(base + j); if (rank_of) rank_of[base + j] = (u32)pos; ++pos; }
        else if (rank_of && base + j < n) rank_of[base + j] = NO_EDGE;
    }
}

// per merged edge: number of k-mers on its path, its last vertex, the bytes its label takes
__global__ __launch_bounds__(BLOCK) void path_measure_kernel(const u32* __restrict__ heads, u64 n_heads, const u64* __restrict__ src,
                                                             const u64* __restrict__ dst, const u64* __restrict__ word, u32 k,
                                                             u32* __restrict__ path_len, u32* __restrict__ label_bytes, u64* __restrict__ end_node,
                                                             u32* __restrict__ keep_node) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n_heads; i += (u64)gridDim.x * BLOCK) {
        const u32 h = heads[i];
        const u64 start = src[h];
        u64 cur = dst[h];
        u32 m = 1;
        while (cur != start) { const u64 w = word[cur]; if (w == NOT_INNER) break; cur = w >> 32; ++m; }
        path_len[i] = m;
        label_bytes[i] = 1 + (k + m - 1 + 3) / 4;
        end_node[i] = cur;
        keep_node[start] = 1; keep_node[cur] = 1;
    }
}

template <int NW>
__global__ __launch_bounds__(BLOCK) void path_write_kernel(const u32* __restrict__ heads, u64 n_heads, const u64* __restrict__ src,
                                                           const u64* __restrict__ dst, const u32* __restrict__ weight, const u64* __restrict__ key,
                                                           const u64* __restrict__ word, const u32* __restrict__ path_len,
                                                           const u64* __restrict__ label_off, const u64* __restrict__ end_node,
                                                           const u32* __restrict__ new_id, u32 k, u64* __restrict__ o_src, u64* __restrict__ o_dst,
                                                           u32* __restrict__ o_weight, uint8_t* __restrict__ o_label) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n_heads; i += (u64)gridDim.x * BLOCK) {
        const u32 h = heads[i], m = path_len[i];
        o_src[i] = new_id[src[h]]; o_dst[i] = new_id[end_node[i]]; o_weight[i] = weight[h];
        const u32 len = k + m - 1;
        uint8_t* out = o_label + label_off[i];
        *out++ = (uint8_t)((4 - len % 4) % 4);                   // compress_edge: padding byte first
        Key<NW> hk;
#pragma unroll
        for (int j = 0; j < NW; ++j) hk.w[j] = key[(u64)h * NW + j];
        u32 acc = 0, have = 0;
        for (u32 j = 0; j < k; ++j) {                             // the head's k bases, most significant first
            acc = (acc << 2) | key_digit(hk, 2 * (k - 1 - j), 2);
            if (++have == 4) { *out++ = (uint8_t)acc; acc = 0; have = 0; }
        }
        u64 cur = dst[h];
        for (u32 s = 1; s < m; ++s) {                             // then the last base of every further edge
            const u64 w = word[cur];
            acc = (acc << 2) | (u32)(key[(u64)(u32)w * NW + NW - 1] & 3);
            if (++have == 4) { *out++ = (uint8_t)acc; acc = 0; have = 0; }
            cur = w >> 32;
        }
        if (have) *out = (uint8_t)(acc << (2 * (4 - have)));     // left-aligned, zero padding in the low bits
    }
}

__global__ __launch_bounds__(BLOCK) void gather_nodes_kernel(const u32* __restrict__ kept, u64 n_kept, const u64* __restrict__ node_key, u32 nw,
                                                             u64* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n_kept; i += (u64)gridDim.x * BLOCK)
        for (u32 w = 0; w < nw; ++w) out[i * nw + w] = node_key[(u64)kept[i] * nw + w];
}

// ---- the exact form (shrink_exact.h decides what is merged and where everything ends up; these write the bytes) -------------------
__global__ __launch_bounds__(BLOCK) void narrow_ids_kernel(const u64* __restrict__ in, u64 n, u32* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) out[i] = (u32)in[i];
}
__global__ __launch_bounds__(BLOCK) void widen_ids_kernel(const u32* __restrict__ in, u64 n, u64* __restrict__ out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) out[i] = in[i];
}
__global__ __launch_bounds__(BLOCK) void age_keys_kernel(const u32* __restrict__ age, u64 n, u64* __restrict__ keys, u32* __restrict__ idx) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) { keys[i] = age[i]; idx[i] = (u32)i; }
}
// per final edge: the original edges on its chain (= k-mers it spells), the bytes its label takes, its weight (its slot's: the first edge's)
__global__ __launch_bounds__(BLOCK) void chain_measure_kernel(const u32* __restrict__ slot, u64 n, const u32* __restrict__ chain_next,
                                                              const u32* __restrict__ weight, u32 k, u32* __restrict__ path_len,
                                                              u32* __restrict__ label_bytes, u32* __restrict__ o_weight) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 h = slot[i];
        u32 m = 1;
        for (u32 c = chain_next[h]; c != NO_EDGE; c = chain_next[c]) ++m;
        path_len[i] = m;
        label_bytes[i] = 1 + (k + m - 1 + 3) / 4;
        o_weight[i] = weight[h];
    }
}
template <int NW>
__global__ __launch_bounds__(BLOCK) void chain_write_kernel(const u32* __restrict__ slot, u64 n, const u32* __restrict__ chain_next,
                                                            const u64* __restrict__ key, const u32* __restrict__ path_len,
                                                            const u64* __restrict__ label_off, u32 k, uint8_t* __restrict__ o_label) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u32 h = slot[i], len = k + path_len[i] - 1;
        uint8_t* out = o_label + label_off[i];
        *out++ = (uint8_t)((4 - len % 4) % 4);                   // compress_edge: padding byte first
        Key<NW> hk;
#pragma unroll
        for (int j = 0; j < NW; ++j) hk.w[j] = key[(u64)h * NW + j];
        u32 acc = 0, have = 0;
        for (u32 j = 0; j < k; ++j) {                             // the first edge's k bases, most significant first
            acc = (acc << 2) | key_digit(hk, 2 * (k - 1 - j), 2);
            if (++have == 4) { *out++ = (uint8_t)acc; acc = 0; have = 0; }
        }
        for (u32 c = chain_next[h]; c != NO_EDGE; c = chain_next[c]) {       // then the last base of every further edge (its remainder)
            acc = (acc << 2) | (u32)(key[(u64)c * NW + NW - 1] & 3);
            if (++have == 4) { *out++ = (uint8_t)acc; acc = 0; have = 0; }
        }
        if (have) *out = (uint8_t)(acc << (2 * (4 - have)));
    }
}

int compact(const u32* flag, u64 n, DevBuf& out_pos, u32* rank_of, u64* n_out, hipStream_t stream) {
    const u64 nblocks = (n + (u64)BLOCK * ITEMS - 1) / ((u64)BLOCK * ITEMS);
    DevBuf counts(stream), offs(stream);
    KCHECK(counts.alloc(nblocks * 4 + 16)); KCHECK(offs.alloc((nblocks + 1) * 8 + 16));
    hipLaunchKernelGGL(flag_count_kernel, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, flag, n, counts.as<u32>());
    KCHECK(dev_scan_counts(counts.as<u32>(), nblocks, offs.as<u64>(), stream));
    KCHECK_HIP(hipMemcpyAsync(n_out, offs.as<u64>() + nblocks, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    KCHECK(out_pos.alloc((*n_out + 1) * 4, stream));
    hipLaunchKernelGGL(flag_write_kernel, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, flag, n, offs.as<u64>(), out_pos.as<u32>(), rank_of);
    KCHECK_HIP(hipGetLastError());
    return KATOME_OK;
}

}  // namespace

int dev_shrink(const ShrinkInput& g, ShrinkOutput& out, hipStream_t stream) {
    const u64 E = g.n_edges, N = g.n_nodes;
    const u32 nw = g.nw, k = g.k;
    out.n_edges = out.n_nodes = out.label_bytes = 0;
    if (E >= 0xFFFFFFFFull || N >= 0xFFFFFFFFull) { set_error("shrink: more than 2^32 edges or nodes on one GPU"); return KATOME_E_UNSUPPORTED; }
    if (E == 0) return KATOME_OK;
    DevBuf node_deg(stream), in_edge(stream), out_edge(stream), flag(stream), covered(stream), heads(stream);
    KCHECK(node_deg.alloc((N + 1) * 8)); KCHECK(in_edge.alloc((N + 1) * 4)); KCHECK(out_edge.alloc((N + 1) * 4));
    KCHECK(flag.alloc((std::max(E, N) + 1) * 4)); KCHECK(covered.alloc((E + 1) * 4));
    KCHECK_HIP(hipMemsetAsync(node_deg.p, 0, N * 8, stream));
    KCHECK_HIP(hipMemsetAsync(covered.p, 0, E * 4, stream));
    const dim3 ge(grid_for(E, BLOCK, 256u * 32u)), blk(BLOCK);
    hipLaunchKernelGGL(adjacency_kernel, ge, blk, 0, stream, g.edge_src, g.edge_dst, E, node_deg.as<u64>(), in_edge.as<u32>(), out_edge.as<u32>());
    DevBuf word(stream);
    KCHECK(word.alloc((N + 1) * 8));
    hipLaunchKernelGGL(node_word_kernel, dim3(grid_for(N, BLOCK, 256u * 32u)), blk, 0, stream, N, node_deg.as<u64>(), in_edge.as<u32>(),
                       out_edge.as<u32>(), g.edge_dst, word.as<u64>());
    node_deg.release(); in_edge.release(); out_edge.release();
    hipLaunchKernelGGL(head_walk_kernel, ge, blk, 0, stream, g.edge_src, g.edge_dst, E, word.as<u64>(), flag.as<u32>(), covered.as<u32>());
    hipLaunchKernelGGL(cycle_head_kernel, ge, blk, 0, stream, g.edge_src, g.edge_dst, E, word.as<u64>(), covered.as<u32>(), flag.as<u32>());
    KCHECK_HIP(hipGetLastError());
    u64 H = 0;
    KCHECK(compact(flag.as<u32>(), E, heads, nullptr, &H, stream));
    covered.release();
    // measure the paths, mark the vertices that stay
    DevBuf path_len(stream), label_bytes(stream), end_node(stream), keep(stream), label_off(stream), kept(stream), new_id(stream);
    KCHECK(path_len.alloc((H + 1) * 4)); KCHECK(label_bytes.alloc((H + 1) * 4)); KCHECK(end_node.alloc((H + 1) * 8));
    KCHECK(keep.alloc((N + 1) * 4)); KCHECK(label_off.alloc((H + 2) * 8)); KCHECK(new_id.alloc((N + 1) * 4));
    KCHECK_HIP(hipMemsetAsync(keep.p, 0, N * 4, stream));
    hipLaunchKernelGGL(path_measure_kernel, dim3(grid_for(H, BLOCK, 256u * 32u)), blk, 0, stream, heads.as<u32>(), H, g.edge_src, g.edge_dst,
                       word.as<u64>(), k, path_len.as<u32>(), label_bytes.as<u32>(),
                       end_node.as<u64>(), keep.as<u32>());
    KCHECK_HIP(hipGetLastError());
    KCHECK(dev_scan_counts(label_bytes.as<u32>(), H, label_off.as<u64>(), stream));
    u64 total_bytes = 0, NK = 0;
    KCHECK_HIP(hipMemcpyAsync(&total_bytes, label_off.as<u64>() + H, 8, hipMemcpyDeviceToHost, stream));
    KCHECK(compact(keep.as<u32>(), N, kept, new_id.as<u32>(), &NK, stream));          // (synchronises: total_bytes is here too)
    // write the result
    KCHECK(out.edge_src.alloc((H + 1) * 8, stream)); KCHECK(out.edge_dst.alloc((H + 1) * 8, stream));
    KCHECK(out.edge_weight.alloc((H + 1) * 4, stream)); KCHECK(out.edge_label.alloc(total_bytes + 16, stream));
    KCHECK(out.node_key.alloc((NK + 1) * 8 * nw, stream));
    if (nw == 1)
        hipLaunchKernelGGL(path_write_kernel<1>, dim3(grid_for(H, BLOCK, 256u * 32u)), blk, 0, stream, heads.as<u32>(), H, g.edge_src, g.edge_dst,
                           g.edge_weight, g.edge_key, word.as<u64>(), path_len.as<u32>(), label_off.as<u64>(), end_node.as<u64>(),
                           new_id.as<u32>(), k, out.edge_src.as<u64>(), out.edge_dst.as<u64>(), out.edge_weight.as<u32>(), out.edge_label.as<uint8_t>());
    else
        hipLaunchKernelGGL(path_write_kernel<2>, dim3(grid_for(H, BLOCK, 256u * 32u)), blk, 0, stream, heads.as<u32>(), H, g.edge_src, g.edge_dst,
                           g.edge_weight, g.edge_key, word.as<u64>(), path_len.as<u32>(), label_off.as<u64>(), end_node.as<u64>(),
                           new_id.as<u32>(), k, out.edge_src.as<u64>(), out.edge_dst.as<u64>(), out.edge_weight.as<u32>(), out.edge_label.as<uint8_t>());
    if (NK) hipLaunchKernelGGL(gather_nodes_kernel, dim3(grid_for(NK, BLOCK, 256u * 32u)), blk, 0, stream, kept.as<u32>(), NK, g.node_key, nw,
                               out.node_key.as<u64>());
    KCHECK_HIP(hipGetLastError());
    {
        const size_t nb = label_off.bytes; out.edge_label_off.adopt(label_off.take(), nb);
        const size_t pb = path_len.bytes; out.edge_kmers.adopt(path_len.take(), pb);
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    out.n_edges = H; out.n_nodes = NK; out.label_bytes = total_bytes;
    return KATOME_OK;
}

// Shrinkable::shrink exactly as the reference computes it (see shrink_exact.h): the device orders the edges by age (petgraph's
// adjacency order) and hands the end points to the sequential statement on the host; what comes back -- for every edge of the shrunk
// graph its slot, its end points under the final numbering, and the chains of original edges behind the slots -- is turned into
// weights, k-mer counts and compress_edge labels on the device again.  edge_age: the index every edge had when it was added (null:
// the edges are still in that order).
int dev_shrink_exact(const ShrinkInput& g, const uint32_t* edge_age, ShrinkOutput& out, double* host_ms, hipStream_t stream) {
    const u64 E = g.n_edges, N = g.n_nodes;
    const u32 nw = g.nw, k = g.k;
    out.n_edges = out.n_nodes = out.label_bytes = 0;
    if (host_ms) *host_ms = 0;
    if (E >= 0xFFFFFFFFull || N >= 0xFFFFFFFFull) { set_error("shrink: more than 2^32 edges or nodes on one GPU"); return KATOME_E_UNSUPPORTED; }
    if (E == 0) return KATOME_OK;
    const dim3 ge(grid_for(E, BLOCK, 256u * 32u)), blk(BLOCK);
    std::vector<uint32_t> h_src, h_dst, h_order;
    try { h_src.resize(E); h_dst.resize(E); if (edge_age) h_order.resize(E); }
    catch (const std::bad_alloc&) { set_error("shrink: out of host memory"); return KATOME_E_OOM; }
    {
        DevBuf s32(stream), d32(stream);
        KCHECK(s32.alloc((E + 1) * 4)); KCHECK(d32.alloc((E + 1) * 4));
        hipLaunchKernelGGL(narrow_ids_kernel, ge, blk, 0, stream, g.edge_src, E, s32.as<u32>());
        hipLaunchKernelGGL(narrow_ids_kernel, ge, blk, 0, stream, g.edge_dst, E, d32.as<u32>());
        KCHECK_HIP(hipGetLastError());
        KCHECK_HIP(hipMemcpyAsync(h_src.data(), s32.p, E * 4, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipMemcpyAsync(h_dst.data(), d32.p, E * 4, hipMemcpyDeviceToHost, stream));
        if (edge_age) {
            DevBuf keys(stream), idx(stream);
            KCHECK(keys.alloc((E + 1) * 8)); KCHECK(idx.alloc((E + 1) * 4));
            hipLaunchKernelGGL(age_keys_kernel, ge, blk, 0, stream, edge_age, E, keys.as<u64>(), idx.as<u32>());
            KCHECK_HIP(hipGetLastError());
            KCHECK(dev_sort(keys.as<u64>(), idx.as<u32>(), E, 1, 32, stream));
            KCHECK_HIP(hipMemcpyAsync(h_order.data(), idx.p, E * 4, hipMemcpyDeviceToHost, stream));
        }
        KCHECK_HIP(hipStreamSynchronize(stream));
    }
    ShrinkExact sx;
    std::vector<uint32_t> kept;
    const auto t0 = std::chrono::steady_clock::now();
    try {
        sx.init(h_src.data(), h_dst.data(), edge_age ? h_order.data() : nullptr, (uint32_t)E, (uint32_t)N);
        std::vector<uint32_t>().swap(h_src); std::vector<uint32_t>().swap(h_dst); std::vector<uint32_t>().swap(h_order);
        sx.run(kept);
    } catch (const std::bad_alloc&) { set_error("shrink: out of host memory"); return KATOME_E_OOM; }
    if (host_ms) *host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const u64 H = sx.n_edges, NK = kept.size();
    DevBuf slot(stream), chain(stream), s32(stream), d32(stream), kept_d(stream), path_len(stream), label_bytes(stream), label_off(stream);
    KCHECK(slot.alloc((H + 1) * 4)); KCHECK(chain.alloc((E + 1) * 4)); KCHECK(s32.alloc((H + 1) * 4)); KCHECK(d32.alloc((H + 1) * 4));
    KCHECK(kept_d.alloc((NK + 1) * 4)); KCHECK(path_len.alloc((H + 1) * 4)); KCHECK(label_bytes.alloc((H + 1) * 4)); KCHECK(label_off.alloc((H + 2) * 8));
    KCHECK_HIP(hipMemcpyAsync(chain.p, sx.chain_next.data(), E * 4, hipMemcpyHostToDevice, stream));
    if (H) {
        KCHECK_HIP(hipMemcpyAsync(slot.p, sx.edge_slot.data(), H * 4, hipMemcpyHostToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(s32.p, sx.edge_node[0].data(), H * 4, hipMemcpyHostToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(d32.p, sx.edge_node[1].data(), H * 4, hipMemcpyHostToDevice, stream));
    }
    if (NK) KCHECK_HIP(hipMemcpyAsync(kept_d.p, kept.data(), NK * 4, hipMemcpyHostToDevice, stream));
    KCHECK(out.edge_src.alloc((H + 1) * 8, stream)); KCHECK(out.edge_dst.alloc((H + 1) * 8, stream));
    KCHECK(out.edge_weight.alloc((H + 1) * 4, stream)); KCHECK(out.node_key.alloc((NK + 1) * 8 * nw, stream));
    const dim3 gh(grid_for(std::max<u64>(H, 1), BLOCK, 256u * 32u));
    u64 total_bytes = 0;
    if (H) {
        hipLaunchKernelGGL(widen_ids_kernel, gh, blk, 0, stream, s32.as<u32>(), H, out.edge_src.as<u64>());
        hipLaunchKernelGGL(widen_ids_kernel, gh, blk, 0, stream, d32.as<u32>(), H, out.edge_dst.as<u64>());
        hipLaunchKernelGGL(chain_measure_kernel, gh, blk, 0, stream, slot.as<u32>(), H, chain.as<u32>(), g.edge_weight, k, path_len.as<u32>(),
                           label_bytes.as<u32>(), out.edge_weight.as<u32>());
        KCHECK_HIP(hipGetLastError());
        KCHECK(dev_scan_counts(label_bytes.as<u32>(), H, label_off.as<u64>(), stream));
        KCHECK_HIP(hipMemcpyAsync(&total_bytes, label_off.as<u64>() + H, 8, hipMemcpyDeviceToHost, stream));
    } else {
        KCHECK_HIP(hipMemsetAsync(label_off.p, 0, 8, stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));          // (total_bytes; and the host vectors may go)
    KCHECK(out.edge_label.alloc(total_bytes + 16, stream));
    if (H) {
        if (nw == 1) hipLaunchKernelGGL(chain_write_kernel<1>, gh, blk, 0, stream, slot.as<u32>(), H, chain.as<u32>(), g.edge_key, path_len.as<u32>(),
                                        label_off.as<u64>(), k, out.edge_label.as<uint8_t>());
        else         hipLaunchKernelGGL(chain_write_kernel<2>, gh, blk, 0, stream, slot.as<u32>(), H, chain.as<u32>(), g.edge_key, path_len.as<u32>(),
                                        label_off.as<u64>(), k, out.edge_label.as<uint8_t>());
    }
    if (NK) hipLaunchKernelGGL(gather_nodes_kernel, dim3(grid_for(NK, BLOCK, 256u * 32u)), blk, 0, stream, kept_d.as<u32>(), NK, g.node_key, nw,
                               out.node_key.as<u64>());
    KCHECK_HIP(hipGetLastError());
    {
        const size_t nb = label_off.bytes; out.edge_label_off.adopt(label_off.take(), nb);
        const size_t pb = path_len.bytes; out.edge_kmers.adopt(path_len.take(), pb);
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    out.n_edges = H; out.n_nodes = NK; out.label_bytes = total_bytes;
    return KATOME_OK;
}

}  // namespace katome
