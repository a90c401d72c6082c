// dist_prune.hip -- Prunable::remove_dead_paths (pruner.rs:36-82) on the SHARDED graph, in the reference's numbering.
//
// After katome_dist_finalize (first-seen order) every rank holds the out-edges of the nodes it owns, each edge with its global
// petgraph index, its source as a LOCAL node and its target as (owner rank, local node there).  What the reference does per
// pass (check_dead_path 229-257 from every vertex without incoming edges, then remove_paths 199-217: swap_remove by
// descending index, remove_single_node after every edge) is cut along what depends on what:
//   * the WALKS follow first_edge(Outgoing) -- the live out-edge added last = largest first-seen index, which never changes
//     (swap_remove re-labels edges, petgraph's lists keep their order) -- for fewer than 2k steps.  A node's out-edges, its
//     degrees and its first out-edge are local to its owner; what a walk reads of a node -- its successor, whether it has an
//     out-edge, whether three or more edges come in -- is one word, and ONE WORD PER NODE OF THE WHOLE GRAPH fits every rank's
//     288 GB (C5 in full: 88 GB): the table is replicated, the words a pass changes are all-gathered, and the walks are the
//     one-GPU loop over a table.  Dead walks are walked a second time; every vertex they pass is a mark that goes to the
//     vertex's owner through a directory sharded by node id.  (Without room for the table: 16-byte walkers that hop from
//     owner to owner, one all-to-all per step, <= 2k - 1 steps.)
//   * the two INDEX REPLAYS (which edge / node sits where after every swap_remove) only involve the marked indices and the
//     tail positions that disappear -- O(what the pass removes), not O(graph).  They run on rank 0 with 64-bit positions
//     (prune.hip's scan + pointer jumping, templated on the index width): the ranks send their marked (position, count)
//     pairs, ask for the fate of their candidate edges (marked, or sitting in the vacated tail) and nodes, and apply the
//     answers -- an edge dies with its removal number or moves to a new position; nothing else of the graph moves.
//   * degrees are kept current incrementally (an out-edge's source is local, its target gets a message), "the removal that
//     touched a vertex last speaks for it" decides which removal takes an endpoint with it (prune.hip, death_emit_kernel).
// No gather: the graph stays sharded, and no index is narrower than 64 bits on the wire or in the replays, so a graph of
// more than 2^32 edges (BASELINE config 5 in full: 1.1e10) is in range; a rank's own share stays below 2^32 edges.
#include <chrono>
#include <cstring>

#include "dist_builder.h"

namespace {

#define KLAUNCH(kernel, n, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid_for((n), BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, __VA_ARGS__)
// whole waves stay in the loop together (wave_append votes)
#define WLOOP(i, n) for (u64 i##0 = (u64)blockIdx.x * BLOCK, i = i##0 + threadIdx.x; i##0 < (n); i##0 += (u64)gridDim.x * BLOCK, i = i##0 + threadIdx.x)

constexpr u64 NONE64 = ~0ull;
constexpr u32 NONE32 = 0xFFFFFFFFu;
constexpr u64 LOW56 = (1ull << 56) - 1;

__device__ __forceinline__ u64 wave_append(bool have, unsigned long long* cursor) {
    const u64 mask = __ballot(have);
    if (!mask) return 0;
    const u32 lane = threadIdx.x & 63;
    u64 base = 0;
    const int leader = __ffsll((unsigned long long)mask) - 1;
    if ((int)lane == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + __popcll(mask & (lane ? (~0ull >> (64 - lane)) : 0ull));
}

// The same for a whole workgroup and CA_ITEMS items per thread: ONE cursor atomic per 2048 items.  (A million waves adding to one
// address take milliseconds -- same-address atomics serialise --, which is what the kernels that look at every edge or node
// of the rank cost while they appended wave by wave: 4-8 ms each per pass, measured, for 0.3 ms of memory traffic.)
// Every thread of the workgroup calls it, the same number of times; returns where this thread's first item goes.
constexpr int CA_ITEMS = 8;
__device__ __forceinline__ u64 block_append(u32 mine, unsigned long long* cursor) {
    __shared__ u32 ca_wtot[BLOCK / 64];
    __shared__ unsigned long long ca_base;
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 v = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += v; }
    __syncthreads();                                         // (the previous call's readers are done with the shared words)
    if (lane == 63) ca_wtot[wave] = incl;
    __syncthreads();
    u32 woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { if (w < (int)wave) woff += ca_wtot[w]; total += ca_wtot[w]; }
    if (threadIdx.x == 0) ca_base = total ? atomicAdd(cursor, (unsigned long long)total) : 0ull;
    __syncthreads();
    return ca_base + woff + (incl - mine);
}
#define TLOOP(t0, n) for (u64 t0 = (u64)blockIdx.x * BLOCK * CA_ITEMS; t0 < (n); t0 += (u64)gridDim.x * BLOCK * CA_ITEMS)
#define KLAUNCH_T(kernel, n, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid_for((n), BLOCK * CA_ITEMS, 256u * 16u)), dim3(BLOCK), 0, stream, __VA_ARGS__)

// ---- set-up -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void run_start_kernel(const u64* __restrict__ lsrc, u64 E, u64 n_src, u64* __restrict__ start) {
    WLOOP(e, E + 1) {
        if (e < E) { if (e == 0 || lsrc[e - 1] != lsrc[e]) start[lsrc[e]] = e; }
        else if (e == E) start[n_src] = E;
    }
}
__global__ __launch_bounds__(BLOCK) void node_init_kernel(u64 N, u64 n_src, const u64* __restrict__ start, u32* __restrict__ outdeg, u32* __restrict__ indeg,
                                                          u32* __restrict__ last_touch, unsigned char* __restrict__ alive) {
    WLOOP(j, N) if (j < N) { outdeg[j] = j < n_src ? (u32)(start[j + 1] - start[j]) : 0u; indeg[j] = 0; last_touch[j] = 0; alive[j] = 1; }
}
// message to a target's owner: (owner rank << 56) | local index there
__global__ __launch_bounds__(BLOCK) void target_msg_kernel(const u64* __restrict__ drank, const u64* __restrict__ dlocal, u64 E, u64* __restrict__ out) {
    WLOOP(e, E) if (e < E) out[e] = (drank[e] << 56) | dlocal[e];
}
__global__ __launch_bounds__(BLOCK) void indeg_add_kernel(const u64* __restrict__ msg, u64 n, u32* indeg) {
    WLOOP(i, n) if (i < n) atomicAdd(&indeg[(u32)(msg[i] & LOW56)], 1u);
}
// first_edge(Outgoing) of every node with out-edges: the live edge of its run with the largest first-seen index
__global__ __launch_bounds__(BLOCK) void first_out_kernel(u64 n_src, const u64* __restrict__ start, const u64* __restrict__ age,
                                                          const unsigned char* __restrict__ alive_e, u32* __restrict__ first_out) {
    WLOOP(j, n_src) if (j < n_src) {
        u64 best = 0; u32 at = NONE32;
        for (u64 e = start[j]; e < start[j + 1]; ++e)
            if (alive_e[e] && (at == NONE32 || age[e] > best)) { best = age[e]; at = (u32)e; }
        first_out[j] = at;
    }
}

// ---- walks ------------------------------------------------------------------------------------------------------------
// walker: A = (rank of the node it stands on << 56) | (that node's local index << 8) | edges collected so far,
//         B = (start rank << 32) | start node's local index
__global__ __launch_bounds__(BLOCK) void input_kernel(u64 N, u64 my_rank, const u32* __restrict__ indeg, const u32* __restrict__ outdeg,
                                                      const unsigned char* __restrict__ alive, u64* __restrict__ A, u64* __restrict__ B,
                                                      unsigned long long* cursor) {
    WLOOP(j, N) {
        const bool is_in = j < N && alive[j] && indeg[j] == 0 && outdeg[j] > 0;      // Externals: Input (pruner.rs:181-183)
        const u64 at = wave_append(is_in, cursor);
        if (is_in) { A[at] = (my_rank << 56) | (j << 8); B[at] = (my_rank << 32) | j; }
    }
}
// one step of check_dead_path for every walker standing on one of this rank's nodes.  -> the walkers that go on (to the
// owner of the next node) and the verdicts of the walks that end dead: D = (start rank << 56) | (start local << 8) | edges
__global__ __launch_bounds__(BLOCK) void walk_step_kernel(const u64* __restrict__ A, const u64* __restrict__ B, u64 n, u32 arrived, u32 two_k,
                                                          const u32* __restrict__ indeg, const u32* __restrict__ outdeg, const u32* __restrict__ first_out,
                                                          const u64* __restrict__ drank, const u64* __restrict__ dlocal,
                                                          u64* __restrict__ oA, u64* __restrict__ oB, unsigned long long* n_out,
                                                          u64* __restrict__ dead, unsigned long long* n_dead) {
    WLOOP(i, n) {
        bool go = false, die = false;
        u64 a = 0, b = 0, steps = 0, nextA = 0;
        if (i < n) {
            a = A[i]; b = B[i];
            const u32 v = (u32)((a & LOW56) >> 8);
            steps = a & 0xFF;
            if (arrived && indeg[v] >= 3) die = true;                       // neighbors_directed(.., Incoming).nth(2) (pruner.rs:253)
            else if (steps + 1 >= two_k) { /* cnt >= 2K: not a dead path (pruner.rs:235-239) */ }
            else if (outdeg[v] == 0) die = true;                            // no first_edge: the path ends here (248-251)
            else {
                const u32 e = first_out[v];
                go = true;
                nextA = (drank[e] << 56) | (dlocal[e] << 8) | (steps + 1);
            }
        }
        const u64 at = wave_append(go, n_out);
        if (go) { oA[at] = nextA; oB[at] = b; }
        const u64 dt = wave_append(die, n_dead);
        if (die) dead[dt] = ((b >> 32) << 56) | ((b & 0xFFFFFFFFull) << 8) | steps;
    }
}
// the second walk of a dead path: its edges are collected (a count per edge: two walks may share a trunk), `left` more to go
__global__ __launch_bounds__(BLOCK) void mark_step_kernel(const u64* __restrict__ M, u64 n, const u32* __restrict__ first_out,
                                                          const u64* __restrict__ drank, const u64* __restrict__ dlocal, u32* mult,
                                                          u64* __restrict__ oM, unsigned long long* n_out) {
    WLOOP(i, n) {
        bool go = false; u64 next = 0;
        if (i < n) {
            const u64 m = M[i], left = m & 0xFF;
            if (left) {
                const u32 v = (u32)((m & LOW56) >> 8), e = first_out[v];
                atomicAdd(&mult[e], 1u);
                if (left > 1) { go = true; next = (drank[e] << 56) | (dlocal[e] << 8) | (left - 1); }
            }
        }
        const u64 at = wave_append(go, n_out);
        if (go) oM[at] = next;
    }
}

// ---- walks on a REPLICATED successor table ------------------------------------------------------------------------------
// 288 GB of HBM per MI355X: one 8-byte word per node of the WHOLE graph -- successor along the first out-edge, "has an
// out-edge", in-degree saturated at 3, alive -- fits every rank (C3/C4: 12.6 GB; C5 in full: 88 GB), indexed by the node's
// first-seen id, which never changes.  With it a walk is local (the one-GPU kernel's loop over a table), and a pass needs a
// handful of collectives instead of two per walk step: the words that changed are all-gathered (O(what the pass removed)),
// the marks of the dead walks travel to their edges' owners through a directory sharded by node id.
constexpr u64 RW_SUCC = (1ull << 40) - 1, RW_HAS_OUT = 1ull << 40, RW_ALIVE = 1ull << 43;
constexpr u32 RW_INDEG_SHIFT = 41;
__global__ __launch_bounds__(BLOCK) void own_words_kernel(u64 N, u64 n_src, const u32* __restrict__ first_out, const u64* __restrict__ edst,
                                                          const u32* __restrict__ indeg, const unsigned char* __restrict__ alive, u64* __restrict__ out) {
    WLOOP(j, N) if (j < N) {
        u64 w = 0;
        if (alive[j]) {
            const u32 fo = j < n_src ? first_out[j] : NONE32;
            const u32 in = indeg[j] < 3u ? indeg[j] : 3u;
            w = RW_ALIVE | ((u64)in << RW_INDEG_SHIFT) | (fo != NONE32 ? (RW_HAS_OUT | (edst[fo] & RW_SUCC)) : RW_SUCC);
        }
        out[j] = w;
    }
}
// the words that differ from what the other ranks hold: {node id, word}
__global__ __launch_bounds__(BLOCK) void delta_kernel(u64 N, const u64* __restrict__ now, u64* __restrict__ held, const u64* __restrict__ gid,
                                                      u64* __restrict__ out, unsigned long long* cursor) {
    TLOOP(t0, N) {
        u64 w[CA_ITEMS]; u32 mine = 0, have = 0;
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) {
            const u64 j = t0 + (u64)k * BLOCK + threadIdx.x;
            if (j < N) { w[k] = now[j]; if (w[k] != held[j]) { have |= 1u << k; ++mine; } }
        }
        u64 at = block_append(mine, cursor);
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) if (have & (1u << k)) {
            const u64 j = t0 + (u64)k * BLOCK + threadIdx.x;
            out[2 * at] = gid[j]; out[2 * at + 1] = w[k]; held[j] = w[k]; ++at;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void apply_words_kernel(const u64* __restrict__ list, u64 n, u64* __restrict__ table) {
    WLOOP(i, n) if (i < n) table[list[2 * i]] = list[2 * i + 1];
}
__global__ __launch_bounds__(BLOCK) void input_list_kernel(u64 N, const u32* __restrict__ indeg, const u32* __restrict__ outdeg,
                                                           const unsigned char* __restrict__ alive, u32* __restrict__ out, unsigned long long* cursor) {
    TLOOP(t0, N) {
        u32 mine = 0, have = 0;
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) {
            const u64 j = t0 + (u64)k * BLOCK + threadIdx.x;
            if (j < N && alive[j] && indeg[j] == 0 && outdeg[j] > 0) { have |= 1u << k; ++mine; }      // Externals: Input (pruner.rs:181-183)
        }
        u64 at = block_append(mine, cursor);
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) if (have & (1u << k)) out[at++] = (u32)(t0 + (u64)k * BLOCK + threadIdx.x);
    }
}
// check_dead_path (pruner.rs:229-257) from every listed vertex, on the replicated table: len[i] = edges of the dead path, 0 = not dead
__global__ __launch_bounds__(BLOCK) void walk_table_kernel(const u32* __restrict__ list, u64 n, const u64* __restrict__ gid, const u64* __restrict__ W,
                                                           u32 two_k, u32* __restrict__ len, unsigned long long* n_dead) {
    u32 dead = 0;
    WLOOP(i, n) if (i < n) {
        u64 cur = gid[list[i]];
        u32 steps = 0, L = 0;
        for (;;) {
            const u64 w = W[cur];
            if (steps && ((w >> RW_INDEG_SHIFT) & 3u) >= 3u) { L = steps; break; }     // nth(2) of the incoming neighbours (253-255)
            if (steps + 1 >= two_k) break;                                              // cnt >= 2K: kept (235-239)
            if (!(w & RW_HAS_OUT)) { L = steps; break; }                                // no first_edge (248-251)
            cur = w & RW_SUCC;
            ++steps;
        }
        len[i] = L;
        dead += L != 0;
    }
    if (dead) atomicAdd(n_dead, (unsigned long long)dead);
}
// the vertices a dead walk passes (their first out-edges are its edges), each addressed to the directory rank of its id
__global__ __launch_bounds__(BLOCK) void walk_marks_kernel(const u32* __restrict__ list, u64 n, const u64* __restrict__ gid, const u64* __restrict__ W,
                                                           const u32* __restrict__ len, const u64* __restrict__ offs, u64 per_rank, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) {
        const u32 L = len[i];
        u64 cur = gid[list[i]], at = offs[i];
        for (u32 s = 0; s < L; ++s) {
            out[at + s] = ((cur / per_rank) << 56) | cur;
            cur = W[cur] & RW_SUCC;
        }
    }
}
// directory: node id -> (owner rank << 56) | local index there
__global__ __launch_bounds__(BLOCK) void dir_msg_kernel(u64 N, u64 my_rank, const u64* __restrict__ gid, u64 per_rank, u64* __restrict__ A, u64* __restrict__ B) {
    WLOOP(j, N) if (j < N) { A[j] = ((gid[j] / per_rank) << 56) | gid[j]; B[j] = (my_rank << 56) | j; }
}
__global__ __launch_bounds__(BLOCK) void dir_fill_kernel(const u64* __restrict__ A, const u64* __restrict__ B, u64 n, u64 base, u64* __restrict__ dir) {
    WLOOP(i, n) if (i < n) dir[(A[i] & LOW56) - base] = B[i];
}
__global__ __launch_bounds__(BLOCK) void dir_forward_kernel(const u64* __restrict__ marks, u64 n, u64 base, const u64* __restrict__ dir, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[i] = dir[(marks[i] & LOW56) - base];
}
__global__ __launch_bounds__(BLOCK) void mark_nodes_kernel(const u64* __restrict__ msg, u64 n, const u32* __restrict__ first_out, u32* mult) {
    WLOOP(i, n) if (i < n) atomicAdd(&mult[first_out[(u32)(msg[i] & LOW56)]], 1u);
}

// ---- marks, candidates, answers -----------------------------------------------------------------------------------------
// (everything that goes to rank 0 carries rank 0 in the top byte: positions stay below 2^56)
__global__ __launch_bounds__(BLOCK) void marks_kernel(const u32* __restrict__ mult, const u64* __restrict__ pos, u64 E, u64* __restrict__ out_pos,
                                                      u64* __restrict__ out_mult, unsigned long long* cursor, unsigned long long* total) {
    u64 sum = 0;
    TLOOP(t0, E) {
        u32 m[CA_ITEMS], mine = 0;
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) {
            const u64 e = t0 + (u64)k * BLOCK + threadIdx.x;
            m[k] = e < E ? mult[e] : 0u;
            mine += m[k] != 0;
        }
        u64 at = block_append(mine, cursor);
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) if (m[k]) {
            const u64 e = t0 + (u64)k * BLOCK + threadIdx.x;
            out_pos[at] = pos[e]; out_mult[at] = m[k]; sum += m[k]; ++at;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(total, (unsigned long long)sum);
}
__global__ __launch_bounds__(BLOCK) void narrow_kernel(const u64* __restrict__ in, u64 n, u32* __restrict__ out) { WLOOP(i, n) if (i < n) out[i] = (u32)in[i]; }
__global__ __launch_bounds__(BLOCK) void widen_kernel(const u32* __restrict__ in, u64 n, u64* __restrict__ out) { WLOOP(i, n) if (i < n) out[i] = in[i]; }
__global__ __launch_bounds__(BLOCK) void iota32_kernel(u32* __restrict__ out, u64 n) { WLOOP(i, n) if (i < n) out[i] = (u32)i; }
// edges whose fate the replay decides: the marked ones and those in the tail [E_new, E) that disappears
__global__ __launch_bounds__(BLOCK) void edge_cand_kernel(const u32* __restrict__ mult, const u64* __restrict__ pos, const unsigned char* __restrict__ alive,
                                                          u64 E, u64 E_new, u64* __restrict__ q, u32* __restrict__ who, unsigned long long* cursor) {
    TLOOP(t0, E) {
        u64 p[CA_ITEMS]; u32 mine = 0, have = 0;
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) {
            const u64 e = t0 + (u64)k * BLOCK + threadIdx.x;
            if (e < E && alive[e]) { p[k] = pos[e]; if (mult[e] != 0 || p[k] >= E_new) { have |= 1u << k; ++mine; } }
        }
        u64 at = block_append(mine, cursor);
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) if (have & (1u << k)) { q[at] = p[k]; who[at] = (u32)(t0 + (u64)k * BLOCK + threadIdx.x); ++at; }
    }
}
__global__ __launch_bounds__(BLOCK) void node_cand_kernel(const u64* __restrict__ npos, const unsigned char* __restrict__ alive, u64 N, u64 N_new,
                                                          u64* __restrict__ q, u32* __restrict__ who, unsigned long long* cursor) {
    TLOOP(t0, N) {
        u64 p[CA_ITEMS]; u32 mine = 0, have = 0;
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) {
            const u64 j = t0 + (u64)k * BLOCK + threadIdx.x;
            if (j < N && alive[j]) { p[k] = npos[j]; if (p[k] >= N_new) { have |= 1u << k; ++mine; } }
        }
        u64 at = block_append(mine, cursor);
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) if (have & (1u << k)) { q[at] = p[k]; who[at] = (u32)(t0 + (u64)k * BLOCK + threadIdx.x); ++at; }
    }
}
// rank 0: answer[i] = table value at the query's place in the sorted key list, NONE64 if the query is not a key
__global__ __launch_bounds__(BLOCK) void answer_u32_kernel(const u64* __restrict__ found, u64 n, const u32* __restrict__ val, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[i] = found[i] == NONE64 ? NONE64 : (u64)val[found[i]];
}
__global__ __launch_bounds__(BLOCK) void answer_to_kernel(const u64* __restrict__ found, u64 n, const u32* __restrict__ idx, const u64* __restrict__ to, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[i] = found[i] == NONE64 ? NONE64 : to[idx[found[i]]];
}
// apply the edge answers: removal number (the edge dies: its source's degree and last touch here, a message to its target's
// owner) or new position
__global__ __launch_bounds__(BLOCK) void edge_apply_kernel(const u32* __restrict__ who, const u64* __restrict__ ord, const u64* __restrict__ newpos, u64 n,
                                                           u64 my_rank, u64* pos, unsigned char* alive_e, const u64* __restrict__ lsrc,
                                                           const u64* __restrict__ drank, const u64* __restrict__ dlocal, u32* outdeg, u32* last_touch,
                                                           u32* __restrict__ dead_e, u32* __restrict__ dead_t, u64* __restrict__ msgA, u64* __restrict__ msgB,
                                                           unsigned long long* n_dead) {
    WLOOP(i, n) {
        bool dies = false; u32 e = 0; u64 t = 0;
        if (i < n) {
            e = who[i];
            if (ord[i] != NONE64) { dies = true; t = ord[i]; }
            else if (newpos[i] != NONE64) pos[e] = newpos[i];
        }
        const u64 at = wave_append(dies, n_dead);
        if (dies) {
            alive_e[e] = 0;
            const u32 a = (u32)lsrc[e];
            atomicSub(&outdeg[a], 1u);
            atomicMax(&last_touch[a], (u32)t + 1u);
            const bool self = drank[e] == my_rank && dlocal[e] == (u64)a;     // (the target role never speaks for a self-loop, prune.hip)
            dead_e[at] = e; dead_t[at] = (u32)t;
            msgA[at] = (drank[e] << 56) | dlocal[e];
            msgB[at] = (t + 1) | (self ? (1ull << 32) : 0ull);
        }
    }
}
__global__ __launch_bounds__(BLOCK) void target_loss_kernel(const u64* __restrict__ msgA, const u64* __restrict__ msgB, u64 n, u32* indeg, u32* last_touch) {
    WLOOP(i, n) if (i < n) {
        const u32 b = (u32)(msgA[i] & LOW56);
        atomicSub(&indeg[b], 1u);
        atomicMax(&last_touch[b], (u32)(msgB[i] & 0xFFFFFFFFull));
    }
}
// which endpoints a removal takes with it: the removal that touched a node last speaks for it (source role here ...
__global__ __launch_bounds__(BLOCK) void die_src_kernel(const u32* __restrict__ dead_e, const u32* __restrict__ dead_t, u64 n, const u64* __restrict__ lsrc,
                                                        const u32* __restrict__ indeg, const u32* __restrict__ outdeg, const u32* __restrict__ last_touch,
                                                        const u64* __restrict__ npos, unsigned char* alive_n, u64* __restrict__ X, u64* __restrict__ Y,
                                                        unsigned long long* cursor) {
    WLOOP(i, n) {
        bool d = false; u32 a = 0; u64 t = 0;
        if (i < n) {
            a = (u32)lsrc[dead_e[i]]; t = dead_t[i];
            d = last_touch[a] == (u32)t + 1u && indeg[a] == 0 && outdeg[a] == 0;
        }
        const u64 at = wave_append(d, cursor);
        if (d) { X[at] = 2 * t; Y[at] = npos[a]; alive_n[a] = 0; }
    }
}
// ... target role at the target's owner)
__global__ __launch_bounds__(BLOCK) void die_dst_kernel(const u64* __restrict__ msgA, const u64* __restrict__ msgB, u64 n, const u32* __restrict__ indeg,
                                                        const u32* __restrict__ outdeg, const u32* __restrict__ last_touch, const u64* __restrict__ npos,
                                                        unsigned char* alive_n, u64* __restrict__ X, u64* __restrict__ Y, unsigned long long* cursor) {
    WLOOP(i, n) {
        bool d = false; u32 b = 0; u64 t1 = 0;
        if (i < n) {
            b = (u32)(msgA[i] & LOW56); t1 = msgB[i] & 0xFFFFFFFFull;
            d = !(msgB[i] >> 32) && last_touch[b] == (u32)t1 && indeg[b] == 0 && outdeg[b] == 0;
        }
        const u64 at = wave_append(d, cursor);
        if (d) { X[at] = 2 * (t1 - 1) + 1; Y[at] = npos[b]; alive_n[b] = 0; }
    }
}
// the last-touch marks are per pass: cleared where this pass set them, once every removal has spoken
__global__ __launch_bounds__(BLOCK) void untouch_src_kernel(const u32* __restrict__ dead_e, u64 n, const u64* __restrict__ lsrc, u32* last_touch) {
    WLOOP(i, n) if (i < n) last_touch[(u32)lsrc[dead_e[i]]] = 0;
}
__global__ __launch_bounds__(BLOCK) void untouch_dst_kernel(const u64* __restrict__ msgA, u64 n, u32* last_touch) {
    WLOOP(i, n) if (i < n) last_touch[(u32)(msgA[i] & LOW56)] = 0;
}
__global__ __launch_bounds__(BLOCK) void fill64_kernel(u64* __restrict__ p, u64 n, u64 v) { WLOOP(i, n) if (i < n) p[i] = v; }
__global__ __launch_bounds__(BLOCK) void scatter64_kernel(const u64* __restrict__ at, const u64* __restrict__ val, u64 n, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[at[i]] = val[i];
}
__global__ __launch_bounds__(BLOCK) void node_apply_kernel(const u32* __restrict__ who, const u64* __restrict__ newpos, u64 n, u64* npos) {
    WLOOP(i, n) if (i < n && newpos[i] != NONE64) npos[who[i]] = newpos[i];
}
__global__ __launch_bounds__(BLOCK) void clear_marks_kernel(const u32* __restrict__ who, u64 n, u32* mult) { WLOOP(i, n) if (i < n) mult[who[i]] = 0; }

// ---- the result -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void alive_list_kernel(const unsigned char* __restrict__ alive, u64 n, u32* __restrict__ out, unsigned long long* cursor) {
    TLOOP(t0, n) {
        u32 mine = 0, have = 0;
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) {
            const u64 i = t0 + (u64)k * BLOCK + threadIdx.x;
            if (i < n && alive[i]) { have |= 1u << k; ++mine; }
        }
        u64 at = block_append(mine, cursor);
#pragma unroll
        for (int k = 0; k < CA_ITEMS; ++k) if (have & (1u << k)) out[at++] = (u32)(t0 + (u64)k * BLOCK + threadIdx.x);
    }
}
__global__ __launch_bounds__(BLOCK) void npos_of_kernel(const u64* __restrict__ msg, u64 n, const u64* __restrict__ npos, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[i] = npos[(u32)(msg[i] & LOW56)];
}
__global__ __launch_bounds__(BLOCK) void gather_src_kernel(const u32* __restrict__ keep, u64 n, const u64* __restrict__ lsrc, const u64* __restrict__ npos, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[i] = npos[(u32)lsrc[keep[i]]];
}

__global__ __launch_bounds__(BLOCK) void scatter_by_idx_kernel(const u64* __restrict__ vals, const u32* __restrict__ at, u64 n, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[at[i]] = vals[i];
}
template <class T>
__global__ __launch_bounds__(BLOCK) void gather_by_kernel(const T* __restrict__ src, const u32* __restrict__ keep, u64 n, T* __restrict__ out) {
    WLOOP(i, n) if (i < n) out[i] = src[keep[i]];
}
template <int NW>
__global__ __launch_bounds__(BLOCK) void gather_keys_by_kernel(const u64* __restrict__ src, const u32* __restrict__ keep, u64 n, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) {
#pragma unroll
        for (int q = 0; q < NW; ++q) out[i * NW + q] = src[(u64)keep[i] * NW + q];
    }
}

// two columns as one 16-byte record per element (one exchange instead of two), and back
__global__ __launch_bounds__(BLOCK) void zip2_kernel(const u64* __restrict__ pa, const u64* __restrict__ B, const u32* __restrict__ pidx, u64 n, u64* __restrict__ out) {
    WLOOP(i, n) if (i < n) { out[2 * i] = pa[i]; out[2 * i + 1] = B[pidx[i]]; }
}
__global__ __launch_bounds__(BLOCK) void unzip2_kernel(const u64* __restrict__ in, u64 n, u64* __restrict__ a, u64* __restrict__ b) {
    WLOOP(i, n) if (i < n) { a[i] = in[2 * i]; b[i] = in[2 * i + 1]; }
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// records with their destination rank in the top byte of column A [and a second column] -> their destinations
struct Routed {
    DevBuf a, b, pidx;
    std::vector<uint64_t> counts, rcnt;
    uint64_t n = 0, n_sent = 0, pair_max = 0;      // pair_max: the largest (rank -> peer) count of the whole exchange
    uint64_t moved = 0;                            // records on the move between ANY two ranks in this exchange
    explicit Routed(hipStream_t s) : a(s), b(s), pidx(s) {}
};
struct Router {
    katome_dist_builder* d; hipStream_t stream; DevBuf bounds;
    double t_part = 0, t_counts = 0, t_xchg = 0, t_gather = 0; uint64_t n_send = 0, n_reply = 0;      // host wall time per kind of step (KATOME_DIST_PRUNE_TRACE)
    Router(katome_dist_builder* d_, hipStream_t s) : d(d_), stream(s), bounds(s) {}
    int init() {
        const int world = d->world();
        std::vector<uint64_t> h(std::max(world - 1, 1), ~0ull);
        for (int p = 0; p + 1 < world; ++p) h[p] = (uint64_t)(p + 1) << 56;
        KCHECK(bounds.alloc(h.size() * 8));
        KCHECK_HIP(hipMemcpyAsync(bounds.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        return KATOME_OK;
    }
    int send(const u64* A, const u64* B, uint64_t n, Routed& out) {
        const int world = d->world();
        out.counts.assign(world, 0); out.rcnt.assign(world, 0); out.n_sent = n;
        DevBuf idx(stream), pa(stream), zipped(stream), landed(stream);
        KCHECK(idx.alloc((n + 1) * 4)); KCHECK(pa.alloc((n + 1) * 8)); KCHECK(out.pidx.alloc((n + 1) * 4));
        ++n_send;
        double t0 = now_ms();
        if (n) {
            KCHECK(dev_iota(idx.as<u32>(), n, stream));
            KCHECK(dev_partition_range(A, idx.as<u32>(), n, bounds.as<u64>(), (uint32_t)world, pa.as<u64>(), out.pidx.as<u32>(), out.counts.data(), stream));
        }
        t_part += now_ms() - t0; t0 = now_ms();
        KCHECK(d->comm->exchange_counts(out.counts.data(), out.rcnt.data(), &out.pair_max, &out.moved));
        t_counts += now_ms() - t0; t0 = now_ms();
        out.n = 0;
        for (uint64_t c : out.rcnt) out.n += c;
        KCHECK(out.a.alloc((out.n + 1) * 8));
        if (B) KCHECK(out.b.alloc((out.n + 1) * 8));
        if (out.moved == 0) return KATOME_OK;                          // nothing travels anywhere: no exchange at all
        if (!B) {
            KCHECK(d->xchg(X_PRUNE, pa.p, out.counts.data(), out.a.p, out.rcnt.data(), 8, stream, false, out.pair_max));
        } else {                                                        // both columns in one exchange of 16-byte records
            KCHECK(zipped.alloc((n + 1) * 16)); KCHECK(landed.alloc((out.n + 1) * 16));
            if (n) KLAUNCH(zip2_kernel, n, stream, pa.as<u64>(), B, out.pidx.as<u32>(), n, zipped.as<u64>());
            KCHECK(d->xchg(X_PRUNE, zipped.p, out.counts.data(), landed.p, out.rcnt.data(), 16, stream, false, out.pair_max));
            if (out.n) KLAUNCH(unzip2_kernel, out.n, stream, landed.as<u64>(), out.n, out.a.as<u64>(), out.b.as<u64>());
            KCHECK_HIP(hipGetLastError());
        }
        KCHECK_HIP(hipStreamSynchronize(stream));
        t_xchg += now_ms() - t0;
        return KATOME_OK;
    }
    // every rank's records (n_mine elements of `elem` bytes) to every rank, in rank order: the same send buffer for all peers
    int allgather(const void* mine, uint64_t n_mine, size_t elem, DevBuf& out, uint64_t* total) {
        const int world = d->world();
        std::vector<uint64_t> all(world, 0);
        KCHECK(d->comm->allgather(n_mine, all.data()));
        uint64_t sum = 0, biggest = 0;
        std::vector<uint64_t> roff(world, 0);
        for (int p = 0; p < world; ++p) { roff[p] = sum; sum += all[p]; biggest = std::max(biggest, all[p]); }
        *total = sum;
        KCHECK(out.alloc((sum + 1) * elem));
        if (sum == 0) return KATOME_OK;
        const uint64_t chunk = std::max<uint64_t>(1, d->comm->max_message_bytes / elem);
        std::vector<uint64_t> so(world), sc(world), ro(world), rc(world);
        for (uint64_t done = 0; done < biggest; done += chunk) {
            for (int p = 0; p < world; ++p) {
                so[p] = std::min(n_mine, done); sc[p] = std::min(n_mine - so[p], chunk);
                const uint64_t rb = std::min(all[p], done);
                ro[p] = roff[p] + rb; rc[p] = std::min(all[p] - rb, chunk);
            }
            KCHECK(d->comm->t->alltoallv(mine, so.data(), sc.data(), out.p, ro.data(), rc.data(), elem, 1, stream));
        }
        katome::ExchangeStats& x = d->xstats[X_PRUNE];
        x.calls += 1; x.bytes_out += n_mine * elem * (uint64_t)(world - 1);
        KCHECK_HIP(hipStreamSynchronize(stream));
        return KATOME_OK;
    }
    // answers aligned with what `r` received travel back; out[i] = the answer to the i-th record of the sender's list
    int reply(const Routed& r, const u64* ans, u64* out) {
        ++n_reply;
        const double t0 = now_ms();
        DevBuf back(stream);
        KCHECK(back.alloc((r.n_sent + 1) * 8));
        KCHECK(d->xchg(X_PRUNE, ans, r.rcnt.data(), back.p, r.counts.data(), 8, stream, false, r.pair_max));
        if (r.n_sent) KLAUNCH(scatter_by_idx_kernel, r.n_sent, stream, back.as<u64>(), r.pidx.as<u32>(), r.n_sent, out);
        KCHECK_HIP(hipGetLastError());
        KCHECK_HIP(hipStreamSynchronize(stream));
        t_xchg += now_ms() - t0;
        return KATOME_OK;
    }
};

// KATOME_DIST_PRUNE_FAIL=edges|nodes: rank 0's replay of that kind reports a failure in the first pass -- tests of "every rank
// leaves with the same error, nobody is left waiting in a collective"
static bool fail_at(const char* what) {
    const char* at = getenv("KATOME_DIST_PRUNE_FAIL");
    return at && !strcmp(at, what);
}
// what rank 0 could not do, as every rank's return value (rank 0 keeps its own message)
static int root_failed(int rank, int neg_status, const char* what) {
    if (rank != 0) set_error("distributed pruning: rank 0 failed in %s (status %d)", what, -neg_status);
    return -neg_status;
}

}  // namespace

extern "C" int katome_dist_remove_dead_paths(katome_dist_builder* d, katome_dist_graph* out, katome_prune_stats* st_out, void* stream_) {
    if (!d) { set_error("null argument"); return KATOME_E_ARG; }
    if (!d->finalized || !d->first_seen) { set_error("katome_dist_remove_dead_paths: a finalized FIRST_SEEN_ORDER build only (petgraph's numbering decides what a repeated index removes)"); return KATOME_E_ARG; }
    if (d->gathered) { set_error("katome_dist_remove_dead_paths: the ranks' shares were gathered (katome_dist_gather); prune the gathered graph on its root"); return KATOME_E_ARG; }
    if (d->dead_paths_removed) {
        // Called again (the reference's pipeline does: asm/basic_assembler.rs:59,73): the first call ran to the fixpoint, nothing has
        // changed the sharded graph since (every other stage works on the gathered one), so the reference's loop would make one pass,
        // find no dead path and leave (pruner.rs:69-72).  Every rank takes this branch: no collective is needed.
        if (st_out) { memset(st_out, 0, sizeof *st_out); st_out->passes = 1; }
        return katome_dist_current_graph(d, out);
    }
    hipStream_t stream = (hipStream_t)stream_;
    katome_builder* b = d->b;
    KCHECK_HIP(hipSetDevice(d->s.device));
    d->comm->use_stream(stream);
    const int rank = d->rank(), world = d->world();
    const uint32_t nw = d->nw, k = d->s.k, two_k = 2 * k;
    const uint64_t E = d->n_edges, N = d->n_nodes, n_src = d->n_src;
    const double t_begin = now_ms();
    katome_prune_stats st;
    memset(&st, 0, sizeof st);
    if (two_k > 255) { set_error("k too large for the walker's step counter"); return KATOME_E_UNSUPPORTED; }       // (the same on every rank)
    {   // a rank whose share cannot be pruned says so to all of them: nobody may leave alone before a collective the others wait in
        uint64_t bad = (E >= 0xFFFFFFFFull || N >= 0xFFFFFFFFull) ? 1 : (E && (!d->edge_lsrc.p || !d->edge_drank.p || !d->edge_dlocal.p)) ? 2 : 0;
        KCHECK(d->comm->allreduce(&bad, 1, OP_MAX));
        if (bad == 1) { set_error("more than 2^32 edges or nodes on one rank"); return KATOME_E_UNSUPPORTED; }
        if (bad) { set_error("katome_dist_remove_dead_paths: a rank's local source / target indices are gone"); return KATOME_E_ARG; }
    }
    Router router(d, stream);
    KCHECK(router.init());
    const u64* lsrc = d->edge_lsrc.as<u64>(); const u64* drank = d->edge_drank.as<u64>(); const u64* dlocal = d->edge_dlocal.as<u64>();
    // ages: the first-seen index an edge was given never changes; `pos` is where it sits now
    DevBuf age(stream), pos(stream), alive_e(stream), mult(stream), start(stream);
    DevBuf npos(stream), indeg(stream), outdeg(stream), first_out(stream), last_touch(stream), alive_n(stream), cursors(stream);
    KCHECK(age.alloc((E + 1) * 8)); KCHECK(pos.alloc((E + 1) * 8)); KCHECK(alive_e.alloc(E + 16)); KCHECK(mult.alloc((E + 1) * 4));
    KCHECK(start.alloc((n_src + 2) * 8));
    KCHECK(npos.alloc((N + 1) * 8)); KCHECK(indeg.alloc((N + 1) * 4)); KCHECK(outdeg.alloc((N + 1) * 4)); KCHECK(first_out.alloc((n_src + 1) * 4));
    KCHECK(last_touch.alloc((N + 1) * 4)); KCHECK(alive_n.alloc(N + 16)); KCHECK(cursors.alloc(64));
    unsigned long long* cur = cursors.as<unsigned long long>();
    auto reset_cursors = [&]() -> int { KCHECK_HIP(hipMemsetAsync(cursors.p, 0, 64, stream)); return KATOME_OK; };
    auto read_cursors = [&](uint64_t* h, int n) -> int {
        KCHECK_HIP(hipMemcpyAsync(h, cursors.p, 8 * n, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        return KATOME_OK;
    };
    if (E) {
        KCHECK_HIP(hipMemcpyAsync(age.p, d->edge_gid.p, E * 8, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(pos.p, d->edge_gid.p, E * 8, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemsetAsync(alive_e.p, 1, E, stream));
        KCHECK_HIP(hipMemsetAsync(mult.p, 0, E * 4, stream));
    }
    KLAUNCH(run_start_kernel, E + 1, stream, lsrc, E, n_src, start.as<u64>());
    if (N) {
        KCHECK_HIP(hipMemcpyAsync(npos.p, d->node_gid.p, N * 8, hipMemcpyDeviceToDevice, stream));
        KLAUNCH(node_init_kernel, N, stream, N, n_src, start.as<u64>(), outdeg.as<u32>(), indeg.as<u32>(), last_touch.as<u32>(), alive_n.as<unsigned char>());
    }
    KCHECK_HIP(hipGetLastError());
    {   // in-degrees: one message per edge to its target's owner (once; kept current afterwards)
        DevBuf msg(stream);
        KCHECK(msg.alloc((E + 1) * 8));
        if (E) KLAUNCH(target_msg_kernel, E, stream, drank, dlocal, E, msg.as<u64>());
        Routed r(stream);
        KCHECK(router.send(msg.as<u64>(), nullptr, E, r));
        if (r.n) KLAUNCH(indeg_add_kernel, r.n, stream, r.a.as<u64>(), r.n, indeg.as<u32>());
        KCHECK_HIP(hipGetLastError());
    }
    uint64_t TE = d->total_edges, TN = d->total_nodes;
    // The walks: on a replicated successor table when every rank has room for one word per node of the whole graph (the
    // default), else walkers that hop from owner to owner (KATOME_DIST_PRUNE_WALKS=table|hop overrides; all ranks agree)
    const uint64_t TN0 = d->total_nodes, per_rank = std::max<uint64_t>(1, (TN0 + world - 1) / world);
    const u64* edst = d->edge_dst.as<u64>();                      // every edge's target as a first-seen node id (the pass installs new arrays at the end)
    bool use_table = true;
    {
        size_t free_b = 0, total_b = 0;
        KCHECK_HIP(hipMemGetInfo(&free_b, &total_b));
        free_b += dev_cached_bytes();
        uint64_t ok = (TN0 + 1) * 8 + (N + 1) * 24 + per_rank * 8 + (4ull << 30) < free_b ? 1 : 0;
        if (const char* e = getenv("KATOME_DIST_PRUNE_WALKS")) ok = strcmp(e, "hop") == 0 ? 0 : strcmp(e, "table") == 0 ? 1 : ok;
        KCHECK(d->comm->allreduce(&ok, 1, OP_MIN));
        use_table = ok != 0;
    }
    DevBuf w_table(stream), w_now(stream), w_held(stream), ngid0(stream), dir(stream);
    if (use_table) {
        KCHECK(w_table.alloc((TN0 + 1) * 8)); KCHECK(w_now.alloc((N + 1) * 8)); KCHECK(w_held.alloc((N + 1) * 8)); KCHECK(ngid0.alloc((N + 1) * 8));
        KCHECK(dir.alloc((per_rank + 1) * 8));
        KCHECK_HIP(hipMemsetAsync(w_table.p, 0, (TN0 + 1) * 8, stream));
        KCHECK_HIP(hipMemsetAsync(w_held.p, 0, (N + 1) * 8, stream));             // (no live node's word is 0: the first pass sends them all)
        if (N) KCHECK_HIP(hipMemcpyAsync(ngid0.p, d->node_gid.p, N * 8, hipMemcpyDeviceToDevice, stream));
        // directory, sharded by node id: where each node lives
        DevBuf da(stream), db(stream);
        KCHECK(da.alloc((N + 1) * 8)); KCHECK(db.alloc((N + 1) * 8));
        if (N) KLAUNCH(dir_msg_kernel, N, stream, N, (u64)rank, ngid0.as<u64>(), per_rank, da.as<u64>(), db.as<u64>());
        Routed r(stream);
        KCHECK(router.send(da.as<u64>(), db.as<u64>(), N, r));
        if (r.n) KLAUNCH(dir_fill_kernel, r.n, stream, r.a.as<u64>(), r.b.as<u64>(), r.n, (u64)rank * per_rank, dir.as<u64>());
        KCHECK_HIP(hipGetLastError());
        KCHECK_HIP(hipStreamSynchronize(stream));
    }
    // positions (of edges and of nodes alike) fit this many bits: what the sorts and look-ups at rank 0 work on
    uint32_t pos_bits = 1;
    while (pos_bits < 64 && ((std::max(d->total_edges, d->total_nodes) + 1) >> pos_bits)) ++pos_bits;
    static const bool trace = getenv("KATOME_DIST_PRUNE_TRACE") != nullptr;
    double t_sec[24] = {0}, t_mark = now_ms();
    auto lap = [&](int i) { if (trace) { (void)hipStreamSynchronize(stream); const double t = now_ms(); t_sec[i] += t - t_mark; t_mark = t; } };
    for (;;) {
        ++st.passes;
        lap(7);
        if (n_src) KLAUNCH(first_out_kernel, n_src, stream, n_src, start.as<u64>(), age.as<u64>(), alive_e.as<unsigned char>(), first_out.as<u32>());
        uint64_t h[4] = {0, 0, 0, 0};
        lap(11);
        if (use_table) {
            // ---- the words of this rank's nodes that changed -> every rank's copy of the table -------------------------------
            if (N) KLAUNCH(own_words_kernel, N, stream, N, n_src, first_out.as<u32>(), edst, indeg.as<u32>(), alive_n.as<unsigned char>(), w_now.as<u64>());
            DevBuf delta(stream), all_delta(stream);
            KCHECK(delta.alloc((N + 1) * 16));
            KCHECK(reset_cursors());
            if (N) KLAUNCH_T(delta_kernel, N, stream, N, w_now.as<u64>(), w_held.as<u64>(), ngid0.as<u64>(), delta.as<u64>(), cur);
            KCHECK_HIP(hipGetLastError());
            KCHECK(read_cursors(h, 1));
            lap(12);
            uint64_t n_delta = 0;
            KCHECK(router.allgather(delta.p, h[0], 16, all_delta, &n_delta));
            if (n_delta) KLAUNCH(apply_words_kernel, n_delta, stream, all_delta.as<u64>(), n_delta, w_table.as<u64>());
            delta.release(); all_delta.release();
            lap(13);
            // ---- walks from every vertex without incoming edges: local --------------------------------------------------------
            DevBuf inputs(stream), len(stream), offs(stream), marks(stream);
            KCHECK(inputs.alloc((N + 1) * 4));
            KCHECK(reset_cursors());
            if (N) KLAUNCH_T(input_list_kernel, N, stream, N, indeg.as<u32>(), outdeg.as<u32>(), alive_n.as<unsigned char>(), inputs.as<u32>(), cur);
            KCHECK_HIP(hipGetLastError());
            KCHECK(read_cursors(h, 1));
            const uint64_t n_in = h[0];
            st.walks += n_in;
            lap(14);
            KCHECK(len.alloc((n_in + 1) * 4)); KCHECK(offs.alloc((n_in + 2) * 8));
            KCHECK(reset_cursors());
            if (n_in) KLAUNCH(walk_table_kernel, n_in, stream, inputs.as<u32>(), n_in, ngid0.as<u64>(), w_table.as<u64>(), two_k, len.as<u32>(), cur);
            KCHECK_HIP(hipGetLastError());
            KCHECK(read_cursors(h, 1));
            st.dead_walks += h[0];
            lap(15);
            uint64_t n_marks = 0;
            if (n_in) {
                KCHECK(dev_scan_counts(len.as<u32>(), n_in, offs.as<u64>(), stream));
                KCHECK_HIP(hipMemcpyAsync(&n_marks, offs.as<u64>() + n_in, 8, hipMemcpyDeviceToHost, stream));
                KCHECK_HIP(hipStreamSynchronize(stream));
            }
            KCHECK(marks.alloc((n_marks + 1) * 8));
            if (n_marks) KLAUNCH(walk_marks_kernel, n_in, stream, inputs.as<u32>(), n_in, ngid0.as<u64>(), w_table.as<u64>(), len.as<u32>(), offs.as<u64>(), per_rank, marks.as<u64>());
            KCHECK_HIP(hipGetLastError());
            lap(16);
            // the marks -> the directory rank of each vertex -> its owner, where its first out-edge is counted
            Routed at_dir(stream), at_owner(stream);
            KCHECK(router.send(marks.as<u64>(), nullptr, n_marks, at_dir));
            DevBuf fwd(stream);
            KCHECK(fwd.alloc((at_dir.n + 1) * 8));
            if (at_dir.n) KLAUNCH(dir_forward_kernel, at_dir.n, stream, at_dir.a.as<u64>(), at_dir.n, (u64)rank * per_rank, dir.as<u64>(), fwd.as<u64>());
            KCHECK_HIP(hipGetLastError());
            KCHECK(router.send(fwd.as<u64>(), nullptr, at_dir.n, at_owner));
            if (at_owner.n) KLAUNCH(mark_nodes_kernel, at_owner.n, stream, at_owner.a.as<u64>(), at_owner.n, first_out.as<u32>(), mult.as<u32>());
            KCHECK_HIP(hipGetLastError());
        } else {
        // ---- walks from every vertex without incoming edges ------------------------------------------------------------
        DevBuf A(stream), B(stream), deadv(stream);
        KCHECK(A.alloc((N + 1) * 8)); KCHECK(B.alloc((N + 1) * 8));
        KCHECK(reset_cursors());
        if (N) KLAUNCH(input_kernel, N, stream, N, (u64)rank, indeg.as<u32>(), outdeg.as<u32>(), alive_n.as<unsigned char>(), A.as<u64>(), B.as<u64>(), cur);
        KCHECK_HIP(hipGetLastError());
        KCHECK(read_cursors(h, 1));
        uint64_t n_walk = h[0], n_deadv = 0, deadv_cap = 0;
        st.walks += n_walk;
        const u64 *wA = A.as<u64>(), *wB = B.as<u64>();
        Routed arrivals(stream);
        for (uint32_t step = 0;; ++step) {
            if (step > two_k) { set_error("distributed pruning: a walk did not end within 2k steps"); return KATOME_E_DEVICE; }
            DevBuf oA(stream), oB(stream);
            KCHECK(oA.alloc((n_walk + 1) * 8)); KCHECK(oB.alloc((n_walk + 1) * 8));
            if (n_deadv + n_walk > deadv_cap) {                              // the verdicts collected on this rank so far + this step's
                DevBuf bigger(stream);
                deadv_cap = (n_deadv + n_walk) * 2 + 1024;
                KCHECK(bigger.alloc(deadv_cap * 8));
                if (n_deadv) KCHECK_HIP(hipMemcpyAsync(bigger.p, deadv.p, n_deadv * 8, hipMemcpyDeviceToDevice, stream));
                const size_t bytes = bigger.bytes;
                deadv.adopt(bigger.take(), bytes);
            }
            h[0] = h[1] = 0;
            if (n_walk) {
                KCHECK(reset_cursors());
                KLAUNCH(walk_step_kernel, n_walk, stream, wA, wB, n_walk, step ? 1u : 0u, two_k, indeg.as<u32>(), outdeg.as<u32>(), first_out.as<u32>(),
                        drank, dlocal, oA.as<u64>(), oB.as<u64>(), cur, deadv.as<u64>() + n_deadv, cur + 1);
                KCHECK_HIP(hipGetLastError());
                KCHECK(read_cursors(h, 2));
            }
            n_deadv += h[1];
            // (the count matrix of the exchange tells every rank whether any walker moved anywhere: no agreement round of its own)
            KCHECK(router.send(oA.as<u64>(), oB.as<u64>(), h[0], arrivals));
            wA = arrivals.a.as<u64>(); wB = arrivals.b.as<u64>(); n_walk = arrivals.n;
            if (arrivals.moved == 0) break;
        }
        st.dead_walks += n_deadv;
        // ---- the dead walks again, from their start vertices, to collect their edges -----------------------------------
        {
            Routed m(stream);
            KCHECK(router.send(deadv.as<u64>(), nullptr, n_deadv, m));
            const u64* wM = m.a.as<u64>();
            uint64_t n_m = m.n;
            Routed arrived(stream);
            for (uint32_t step = 0;; ++step) {
                if (step > two_k) { set_error("distributed pruning: a marking walk did not end within 2k steps"); return KATOME_E_DEVICE; }
                DevBuf oM(stream);
                KCHECK(oM.alloc((n_m + 1) * 8));
                h[0] = 0;
                if (n_m) {
                    KCHECK(reset_cursors());
                    KLAUNCH(mark_step_kernel, n_m, stream, wM, n_m, first_out.as<u32>(), drank, dlocal, mult.as<u32>(), oM.as<u64>(), cur);
                    KCHECK_HIP(hipGetLastError());
                    KCHECK(read_cursors(h, 1));                   // (synchronises: the list just walked may be given up below)
                }
                KCHECK(router.send(oM.as<u64>(), nullptr, h[0], arrived));
                wM = arrived.a.as<u64>(); n_m = arrived.n;
                if (arrived.moved == 0) break;
            }
        }
        }
        lap(0);
        // ---- marked (position, count) pairs -> rank 0, which replays remove_paths' swap_removes --------------------------
        DevBuf mp(stream), mm(stream);
        KCHECK(mp.alloc((E + 1) * 8)); KCHECK(mm.alloc((E + 1) * 8));
        KCHECK(reset_cursors());
        if (E) KLAUNCH_T(marks_kernel, E, stream, mult.as<u32>(), pos.as<u64>(), E, mp.as<u64>(), mm.as<u64>(), cur, cur + 1);
        KCHECK_HIP(hipGetLastError());
        KCHECK(read_cursors(h, 2));
        const uint64_t u_local = h[0];
        st.marked += h[1];
        uint64_t u_total = u_local;
        KCHECK(d->comm->allreduce(&u_total, 1, OP_SUM));
        if (u_total == 0) break;                                            // "if to_remove.is_empty() ... Graph is pruned" (pruner.rs:69-72)
        lap(8);
        Routed marks(stream);
        KCHECK(router.send(mp.as<u64>(), mm.as<u64>(), u_local, marks));   // (top byte 0: everything goes to rank 0)
        lap(9);
        mp.release(); mm.release();
        // rank 0's tables for the questions that follow
        DevBuf victims(stream), to_e(stream), from_e(stream), vict_sorted(stream), vict_ord(stream), from_sorted(stream), from_idx(stream);
        ReplayScratch sc(stream);
        uint64_t m_removed = 0, E_new = TE, n_moves = 0, dups = 0;
        // (what fails on rank 0 alone must not leave the others waiting in the next collective: its status travels with the counts
        // every rank agrees on below, and all of them return it)
        auto root_edge_replay = [&]() -> int {
            const uint64_t u = marks.n;
            DevBuf m32(stream);
            KCHECK(m32.alloc((u + 1) * 4));
            KLAUNCH(narrow_kernel, u, stream, marks.b.as<u64>(), u, m32.as<u32>());
            KCHECK(dev_sort(marks.a.as<u64>(), m32.as<u32>(), u, 1, pos_bits, stream));
            KCHECK(dev_replay_edges64(marks.a.as<u64>(), m32.as<u32>(), u, TE, sc, victims, to_e, from_e, &m_removed, &E_new, &n_moves, &dups, stream));
            lap(10);
            if (m_removed >= 0xFFFFFFFFull) { set_error("more than 2^32 edges removed in one pass"); return KATOME_E_UNSUPPORTED; }
            KCHECK(vict_sorted.alloc((m_removed + 1) * 8)); KCHECK(vict_ord.alloc((m_removed + 1) * 4));
            if (m_removed) {
                KCHECK_HIP(hipMemcpyAsync(vict_sorted.p, victims.p, m_removed * 8, hipMemcpyDeviceToDevice, stream));
                KLAUNCH(iota32_kernel, m_removed, stream, vict_ord.as<u32>(), m_removed);
                KCHECK(dev_sort(vict_sorted.as<u64>(), vict_ord.as<u32>(), m_removed, 1, pos_bits, stream));
            }
            KCHECK(from_sorted.alloc((n_moves + 1) * 8)); KCHECK(from_idx.alloc((n_moves + 1) * 4));
            if (n_moves) {
                KCHECK_HIP(hipMemcpyAsync(from_sorted.p, from_e.p, n_moves * 8, hipMemcpyDeviceToDevice, stream));
                KLAUNCH(iota32_kernel, n_moves, stream, from_idx.as<u32>(), n_moves);
                KCHECK(dev_sort(from_sorted.as<u64>(), from_idx.as<u32>(), n_moves, 1, pos_bits, stream));
            }
            KCHECK_HIP(hipGetLastError());
            return KATOME_OK;
        };
        int root_rc = rank == 0 ? root_edge_replay() : KATOME_OK;
        if (rank == 0 && root_rc == KATOME_OK && fail_at("edges")) { set_error("distributed pruning: KATOME_DIST_PRUNE_FAIL=edges"); root_rc = KATOME_E_UNSUPPORTED; }
        lap(1);
        uint64_t agreed[3] = {rank == 0 && root_rc == KATOME_OK ? TE - E_new : 0, rank == 0 ? dups : 0, (uint64_t)(-root_rc)};
        KCHECK(d->comm->allreduce(agreed, 3, OP_MAX));
        if (agreed[2]) return root_failed(rank, (int)agreed[2], "the replay of the removed edges");
        m_removed = agreed[0]; E_new = TE - m_removed;
        st.removed_edges += m_removed; st.removed_by_duplicates += agreed[1];
        // ---- every rank asks for the fate of its candidate edges -----------------------------------------------------------
        DevBuf q(stream), who(stream), ans_ord(stream), ans_pos(stream);
        KCHECK(q.alloc((E + 1) * 8)); KCHECK(who.alloc((E + 1) * 4));
        KCHECK(reset_cursors());
        if (E) KLAUNCH_T(edge_cand_kernel, E, stream, mult.as<u32>(), pos.as<u64>(), alive_e.as<unsigned char>(), E, E_new, q.as<u64>(), who.as<u32>(), cur);
        KCHECK_HIP(hipGetLastError());
        KCHECK(read_cursors(h, 1));
        const uint64_t n_q = h[0];
        KCHECK(ans_ord.alloc((n_q + 1) * 8)); KCHECK(ans_pos.alloc((n_q + 1) * 8));
        {
            Routed asked(stream);
            KCHECK(router.send(q.as<u64>(), nullptr, n_q, asked));
            DevBuf found(stream), a1(stream), a2(stream);
            KCHECK(found.alloc((asked.n + 1) * 8)); KCHECK(a1.alloc((asked.n + 1) * 8)); KCHECK(a2.alloc((asked.n + 1) * 8));
            if (asked.n) {                                                   // (rank 0 only)
                if (m_removed) KCHECK(dev_rank(vict_sorted.as<u64>(), m_removed, 1, pos_bits, asked.a.as<u64>(), asked.n, found.as<u64>(), stream));
                else KCHECK_HIP(hipMemsetAsync(found.p, 0xFF, asked.n * 8, stream));
                KLAUNCH(answer_u32_kernel, asked.n, stream, found.as<u64>(), asked.n, vict_ord.as<u32>(), a1.as<u64>());
                if (n_moves) KCHECK(dev_rank(from_sorted.as<u64>(), n_moves, 1, pos_bits, asked.a.as<u64>(), asked.n, found.as<u64>(), stream));
                else KCHECK_HIP(hipMemsetAsync(found.p, 0xFF, asked.n * 8, stream));
                KLAUNCH(answer_to_kernel, asked.n, stream, found.as<u64>(), asked.n, from_idx.as<u32>(), to_e.as<u64>(), a2.as<u64>());
                KCHECK_HIP(hipGetLastError());
            }
            KCHECK(router.reply(asked, a1.as<u64>(), ans_ord.as<u64>()));
            KCHECK(router.reply(asked, a2.as<u64>(), ans_pos.as<u64>()));
        }
        victims.release(); to_e.release(); from_e.release(); vict_sorted.release(); vict_ord.release(); from_sorted.release(); from_idx.release();
        lap(2);
        // ---- apply: dead edges, their sources' degrees here, messages to their targets' owners ----------------------------
        DevBuf dead_e(stream), dead_t(stream), msgA(stream), msgB(stream);
        KCHECK(dead_e.alloc((n_q + 1) * 4)); KCHECK(dead_t.alloc((n_q + 1) * 4)); KCHECK(msgA.alloc((n_q + 1) * 8)); KCHECK(msgB.alloc((n_q + 1) * 8));
        KCHECK(reset_cursors());
        if (n_q) {
            KLAUNCH(edge_apply_kernel, n_q, stream, who.as<u32>(), ans_ord.as<u64>(), ans_pos.as<u64>(), n_q, (u64)rank, pos.as<u64>(), alive_e.as<unsigned char>(),
                    lsrc, drank, dlocal, outdeg.as<u32>(), last_touch.as<u32>(), dead_e.as<u32>(), dead_t.as<u32>(), msgA.as<u64>(), msgB.as<u64>(), cur);
            KLAUNCH(clear_marks_kernel, n_q, stream, who.as<u32>(), n_q, mult.as<u32>());
        }
        KCHECK_HIP(hipGetLastError());
        KCHECK(read_cursors(h, 1));
        const uint64_t n_dead = h[0];
        Routed losses(stream);
        KCHECK(router.send(msgA.as<u64>(), msgB.as<u64>(), n_dead, losses));
        if (losses.n) KLAUNCH(target_loss_kernel, losses.n, stream, losses.a.as<u64>(), losses.b.as<u64>(), losses.n, indeg.as<u32>(), last_touch.as<u32>());
        KCHECK_HIP(hipGetLastError());
        KCHECK_HIP(hipStreamSynchronize(stream));
        { uint64_t bar = 0; KCHECK(d->comm->allreduce(&bar, 1, OP_MAX)); }   // every rank has applied every loss before anyone decides who dies
        lap(3);
        // ---- which endpoints go with which removal -> rank 0, which replays remove_single_node ------------------------------
        DevBuf X(stream), Y(stream);
        KCHECK(X.alloc((n_dead + losses.n + 1) * 8)); KCHECK(Y.alloc((n_dead + losses.n + 1) * 8));
        KCHECK(reset_cursors());
        if (n_dead) KLAUNCH(die_src_kernel, n_dead, stream, dead_e.as<u32>(), dead_t.as<u32>(), n_dead, lsrc, indeg.as<u32>(), outdeg.as<u32>(), last_touch.as<u32>(),
                            npos.as<u64>(), alive_n.as<unsigned char>(), X.as<u64>(), Y.as<u64>(), cur);
        if (losses.n) KLAUNCH(die_dst_kernel, losses.n, stream, losses.a.as<u64>(), losses.b.as<u64>(), losses.n, indeg.as<u32>(), outdeg.as<u32>(), last_touch.as<u32>(),
                              npos.as<u64>(), alive_n.as<unsigned char>(), X.as<u64>(), Y.as<u64>(), cur);
        if (n_dead) KLAUNCH(untouch_src_kernel, n_dead, stream, dead_e.as<u32>(), n_dead, lsrc, last_touch.as<u32>());
        if (losses.n) KLAUNCH(untouch_dst_kernel, losses.n, stream, losses.a.as<u64>(), losses.n, last_touch.as<u32>());
        KCHECK_HIP(hipGetLastError());
        KCHECK(read_cursors(h, 1));
        Routed dies(stream);
        KCHECK(router.send(X.as<u64>(), Y.as<u64>(), h[0], dies));
        DevBuf to_n(stream), from_n(stream), nfrom_sorted(stream), nfrom_idx(stream);
        uint64_t n_nmoves = 0, N_new = TN;
        auto root_node_replay = [&]() -> int {
            DevBuf die(stream);
            NodeReplayScratch nsc(stream);
            KCHECK(die.alloc((2 * m_removed + 2) * 8));
            KLAUNCH(fill64_kernel, 2 * m_removed, stream, die.as<u64>(), 2 * m_removed, NONE64);
            if (dies.n) KLAUNCH(scatter64_kernel, dies.n, stream, dies.a.as<u64>(), dies.b.as<u64>(), dies.n, die.as<u64>());
            KCHECK_HIP(hipGetLastError());
            int fell_back = 0;      // (1: the pass's node moves chained further than the device form follows and were replayed on the host)
            const double t_replay = now_ms();
            KCHECK(dev_replay_nodes64(die.as<u64>(), m_removed, TN, nsc, to_n, from_n, &n_nmoves, &N_new, &fell_back, stream));
            if (fell_back) st.host_ms += now_ms() - t_replay;
            KCHECK(nfrom_sorted.alloc((n_nmoves + 1) * 8)); KCHECK(nfrom_idx.alloc((n_nmoves + 1) * 4));
            if (n_nmoves) {
                KCHECK_HIP(hipMemcpyAsync(nfrom_sorted.p, from_n.p, n_nmoves * 8, hipMemcpyDeviceToDevice, stream));
                KLAUNCH(iota32_kernel, n_nmoves, stream, nfrom_idx.as<u32>(), n_nmoves);
                KCHECK(dev_sort(nfrom_sorted.as<u64>(), nfrom_idx.as<u32>(), n_nmoves, 1, pos_bits, stream));
            }
            KCHECK_HIP(hipGetLastError());
            return KATOME_OK;
        };
        root_rc = rank == 0 ? root_node_replay() : KATOME_OK;
        if (rank == 0 && root_rc == KATOME_OK && fail_at("nodes")) { set_error("distributed pruning: KATOME_DIST_PRUNE_FAIL=nodes"); root_rc = KATOME_E_UNSUPPORTED; }
        lap(4);
        uint64_t removed_nodes[2] = {rank == 0 && root_rc == KATOME_OK ? TN - N_new : 0, (uint64_t)(-root_rc)};
        KCHECK(d->comm->allreduce(removed_nodes, 2, OP_MAX));
        if (removed_nodes[1]) return root_failed(rank, (int)removed_nodes[1], "the replay of the removed nodes");
        N_new = TN - removed_nodes[0];
        st.removed_nodes += removed_nodes[0];
        {   // the nodes in the vacated tail that stay ask where they go
            DevBuf nq(stream), nwho(stream), nans(stream);
            KCHECK(nq.alloc((N + 1) * 8)); KCHECK(nwho.alloc((N + 1) * 4));
            KCHECK(reset_cursors());
            if (N) KLAUNCH_T(node_cand_kernel, N, stream, npos.as<u64>(), alive_n.as<unsigned char>(), N, N_new, nq.as<u64>(), nwho.as<u32>(), cur);
            KCHECK_HIP(hipGetLastError());
            KCHECK(read_cursors(h, 1));
            const uint64_t n_nq = h[0];
            KCHECK(nans.alloc((n_nq + 1) * 8));
            Routed asked(stream);
            KCHECK(router.send(nq.as<u64>(), nullptr, n_nq, asked));
            DevBuf found(stream), a2(stream);
            KCHECK(found.alloc((asked.n + 1) * 8)); KCHECK(a2.alloc((asked.n + 1) * 8));
            if (asked.n) {
                if (n_nmoves) KCHECK(dev_rank(nfrom_sorted.as<u64>(), n_nmoves, 1, pos_bits, asked.a.as<u64>(), asked.n, found.as<u64>(), stream));
                else KCHECK_HIP(hipMemsetAsync(found.p, 0xFF, asked.n * 8, stream));
                KLAUNCH(answer_to_kernel, asked.n, stream, found.as<u64>(), asked.n, nfrom_idx.as<u32>(), to_n.as<u64>(), a2.as<u64>());
                KCHECK_HIP(hipGetLastError());
            }
            KCHECK(router.reply(asked, a2.as<u64>(), nans.as<u64>()));
            if (n_nq) KLAUNCH(node_apply_kernel, n_nq, stream, nwho.as<u32>(), nans.as<u64>(), n_nq, npos.as<u64>());
            KCHECK_HIP(hipGetLastError());
        }
        TE = E_new; TN = N_new;
        lap(5);
    }
    if (trace)
        fprintf(stderr, "[dist prune] rank %d router: %llu sends, %llu replies; host ms: partition %.1f, count rounds %.1f, exchanges %.1f\n", rank,
                (unsigned long long)router.n_send, (unsigned long long)router.n_reply, router.t_part, router.t_counts, router.t_xchg);
    if (trace)
        fprintf(stderr, "[dist prune] rank %d: %llu passes; ms: walks+marks %.1f, marks->root+edge replay %.1f, edge questions %.1f, apply+losses %.1f, "
                        "dies+node replay %.1f, node questions %.1f; of the second: marks kernel %.1f, send %.1f, sort+replay %.1f; walks: first_out %.1f, words+delta %.1f, allgather+apply %.1f, "
                        "input list %.1f, walk %.1f, scan+marks %.1f, marks routed %.1f\n", rank, (unsigned long long)st.passes, t_sec[0], t_sec[1], t_sec[2], t_sec[3], t_sec[4], t_sec[5], t_sec[8], t_sec[9], t_sec[10],
                t_sec[11], t_sec[12], t_sec[13], t_sec[14], t_sec[15], t_sec[16], t_sec[0]);
    // ---- the pruned graph, still sharded: survivors compacted, positions as indices, ages kept ---------------------------------
    DevBuf keep_e(stream), keep_n(stream);
    KCHECK(keep_e.alloc((E + 1) * 4)); KCHECK(keep_n.alloc((N + 1) * 4));
    KCHECK(reset_cursors());
    if (E) KLAUNCH_T(alive_list_kernel, E, stream, alive_e.as<unsigned char>(), E, keep_e.as<u32>(), cur);
    if (N) KLAUNCH_T(alive_list_kernel, N, stream, alive_n.as<unsigned char>(), N, keep_n.as<u32>(), cur + 1);
    KCHECK_HIP(hipGetLastError());
    uint64_t hc[2] = {0, 0};
    KCHECK(read_cursors(hc, 2));
    const uint64_t E2 = hc[0], N2 = hc[1];
    // (wave_append hands out places in no particular order: back into local order, which is key order)
    {
        DevBuf k64(stream);
        KCHECK(k64.alloc((std::max(E2, N2) + 1) * 8));
        auto order = [&](DevBuf& keep, uint64_t n) -> int {
            if (n < 2) return KATOME_OK;
            KLAUNCH(widen_kernel, n, stream, keep.as<u32>(), n, k64.as<u64>());
            KCHECK(dev_sort(k64.as<u64>(), nullptr, n, 1, 32, stream));
            KLAUNCH(narrow_kernel, n, stream, k64.as<u64>(), n, keep.as<u32>());
            KCHECK_HIP(hipGetLastError());
            return KATOME_OK;
        };
        KCHECK(order(keep_e, E2)); KCHECK(order(keep_n, N2));
    }
    DevBuf n_key(stream), n_gid(stream), e_key(stream), e_w(stream), e_gid(stream), e_age(stream), e_src(stream), e_dst(stream);
    KCHECK(n_key.alloc((N2 + 1) * 8 * nw)); KCHECK(n_gid.alloc((N2 + 1) * 8));
    KCHECK(e_key.alloc((E2 + 1) * 8 * nw)); KCHECK(e_w.alloc((E2 + 1) * 4)); KCHECK(e_gid.alloc((E2 + 1) * 8)); KCHECK(e_age.alloc((E2 + 1) * 8));
    KCHECK(e_src.alloc((E2 + 1) * 8)); KCHECK(e_dst.alloc((E2 + 1) * 8));
    {   // targets' indices: asked of their owners (all edges; the dead ones are dropped below)
        DevBuf msg(stream), dst_all(stream), ans(stream);
        KCHECK(msg.alloc((E + 1) * 8)); KCHECK(dst_all.alloc((E + 1) * 8));
        if (E) KLAUNCH(target_msg_kernel, E, stream, drank, dlocal, E, msg.as<u64>());
        Routed asked(stream);
        KCHECK(router.send(msg.as<u64>(), nullptr, E, asked));
        KCHECK(ans.alloc((asked.n + 1) * 8));
        if (asked.n) KLAUNCH(npos_of_kernel, asked.n, stream, asked.a.as<u64>(), asked.n, npos.as<u64>(), ans.as<u64>());
        KCHECK_HIP(hipGetLastError());
        KCHECK(router.reply(asked, ans.as<u64>(), dst_all.as<u64>()));
        if (E2) KLAUNCH(gather_by_kernel<u64>, E2, stream, dst_all.as<u64>(), keep_e.as<u32>(), E2, e_dst.as<u64>());
    }
    if (E2) {
        KLAUNCH(gather_src_kernel, E2, stream, keep_e.as<u32>(), E2, lsrc, npos.as<u64>(), e_src.as<u64>());
        KLAUNCH(gather_by_kernel<u64>, E2, stream, pos.as<u64>(), keep_e.as<u32>(), E2, e_gid.as<u64>());
        KLAUNCH(gather_by_kernel<u64>, E2, stream, age.as<u64>(), keep_e.as<u32>(), E2, e_age.as<u64>());
        KLAUNCH(gather_by_kernel<u32>, E2, stream, b->edge_weight.as<u32>(), keep_e.as<u32>(), E2, e_w.as<u32>());
        if (nw == 1) KLAUNCH(gather_keys_by_kernel<1>, E2, stream, b->edge_key.as<u64>(), keep_e.as<u32>(), E2, e_key.as<u64>());
        else         KLAUNCH(gather_keys_by_kernel<2>, E2, stream, b->edge_key.as<u64>(), keep_e.as<u32>(), E2, e_key.as<u64>());
    }
    if (N2) {
        KLAUNCH(gather_by_kernel<u64>, N2, stream, npos.as<u64>(), keep_n.as<u32>(), N2, n_gid.as<u64>());
        if (nw == 1) KLAUNCH(gather_keys_by_kernel<1>, N2, stream, d->node_key.as<u64>(), keep_n.as<u32>(), N2, n_key.as<u64>());
        else         KLAUNCH(gather_keys_by_kernel<2>, N2, stream, d->node_key.as<u64>(), keep_n.as<u32>(), N2, n_key.as<u64>());
    }
    KCHECK_HIP(hipGetLastError());
    KCHECK_HIP(hipStreamSynchronize(stream));
    auto install = [&](DevBuf& dst, DevBuf& src) { const size_t bytes = src.bytes; dst.stream = stream; dst.adopt(src.take(), bytes); };
    install(b->edge_key, e_key); install(b->edge_weight, e_w); install(d->edge_src, e_src); install(d->edge_dst, e_dst);
    install(d->edge_gid, e_gid); install(d->edge_age, e_age); install(d->node_key, n_key); install(d->node_gid, n_gid);
    b->edge_seq.release();
    d->edge_lsrc.release(); d->edge_drank.release(); d->edge_dlocal.release();      // (local indices of the unpruned share: stale now)
    const uint32_t lstride = label_stride_for_k(k);
    KCHECK(d->edge_label.alloc((E2 + 1) * (size_t)lstride + 16, stream));
    KCHECK(dev_labels(b->edge_key.as<u64>(), E2, k, d->edge_label.as<uint8_t>(), stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    b->n_edges = E2;
    d->n_edges = E2; d->n_nodes = N2; d->total_edges = TE; d->total_nodes = TN; d->n_src = 0;
    d->dead_paths_removed = true;
    {   // the walks and marks of all ranks
        uint64_t sums[3] = {st.walks, st.dead_walks, st.marked};
        KCHECK(d->comm->allreduce(sums, 3, OP_SUM));
        st.walks = sums[0]; st.dead_walks = sums[1]; st.marked = sums[2];
    }
    st.total_ms = now_ms() - t_begin;
    if (st_out) *st_out = st;
    return katome_dist_current_graph(d, out);
}
