// prune.hip -- Prunable::remove_dead_paths for PtGraph (reference src/katome/algorithms/pruner.rs:36-82, with
// Externals 165-195, remove_paths 199-217, remove_single_node 219-225, check_dead_path 229-257) on the
// first-seen-ordered device graph, index for index.
//
// The reference repeats until nothing is found: (1) from every vertex without incoming edges follow
// first_edge(Outgoing) for fewer than 2k steps; a walk that ends at a vertex without out-edges or at one with three
// or more incoming edges is a dead path (walks from vertices without out-edges can never end that way, so they are
// skipped here); (2) the collected petgraph edge indices are sorted descending and removed ONE BY ONE with
// Graph::remove_edge, each followed by remove_node of the endpoints that became isolated.  petgraph 0.4.13 removes by
// swap_remove, so every removal re-labels the last edge / node, and an index that two walks both collected removes
// whatever edge was swapped in.  That bookkeeping decides the final numbering (and, through the duplicates, the
// surviving set), so it is reproduced exactly:
// everything on the device --
//   * degrees and first_edge(Outgoing) of every vertex (petgraph's list head = the live out-edge added last = largest
//     first-seen index; swap_remove re-labels edges but never reorders the lists), per-vertex edge slots, the list of
//     Input vertices, the walks (after small passes only those within reach of a change), the marks per edge index;
//   * the two swap_remove replays in parallel form (edges: a scan of n -> max(n - c, d) functions + pointer jumping;
//     vertices: where the occupant of each vacated tail position goes, settled in rounds), which endpoints die with
//     which removal, and the resulting moves applied to the arrays;
//   * prune_replay.h keeps the sequential statement of the two replays for a pass whose vertex moves chain further than
//     the device form follows (and for A/B runs).
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "common.h"
#include "prune_replay.h"

namespace katome {
namespace {

typedef uint32_t u32;
constexpr u32 NONE32 = REPLAY_NONE;
constexpr int MARK_ITEMS = 8;          // edges per thread in the mark compaction
static_assert(BLOCK * MARK_ITEMS == 2048, "one 'touched' byte per workgroup of the mark compaction (MARK_GROUP_SHIFT)");
// node_deg[v], one word per node so that a step of a walk is one look-up: bits 0-15 in-degree, bits 16-29 out-degree (a
// (k-1)-mer has at most four edges either way), bit 30 "listed as an Input vertex", bit 31 "lost its first out-edge, first_out is being rebuilt", bits 32-63
// the node its first out-edge leads to (NONE32: no out-edge).  first_out[v]: (first-seen index + 1) << 32 | position of
// the out-edge with the largest first-seen index (0 = none).  Both are built once and then kept up to date by the
// kernels below as edges go and as edges and nodes are moved.
constexpr u64 IN_ONE = 1ull, OUT_ONE = 1ull << 16, IN_MASK = 0xFFFFull, DEG_MASK = 0x7FFFFFFFull;
constexpr u64 REDO_FIRST_OUT = 1ull << 31;
constexpr u64 LISTED = 1ull << 30;          // the vertex is in the list of Input vertices that is carried from pass to pass
constexpr u64 DEGREES = 0x3FFFFFFFull;      // in- and out-degree
constexpr u32 DIED = 0xFFFFFFFFu;           // in last_touch: the vertex is removed in this pass
__global__ __launch_bounds__(BLOCK) void degree_kernel(const u64* __restrict__ src, const u64* __restrict__ dst,
                                                       const u32* __restrict__ orig, u64 E, u64* __restrict__ node_deg,
                                                       u64* __restrict__ first_out) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        const u64 a = src[e], b = dst[e];
        atomicAdd((unsigned long long*)&node_deg[a], OUT_ONE);
        atomicAdd((unsigned long long*)&node_deg[b], IN_ONE);
        atomicMax((unsigned long long*)&first_out[a], ((unsigned long long)(orig[e] + 1u) << 32) | (unsigned long long)e);
    }
}
__device__ __forceinline__ u64 with_successor(u64 deg, u64 fo, const u64* __restrict__ dst) {
    return (deg & DEG_MASK) | ((fo ? dst[(u32)fo] : (u64)NONE32) << 32);
}
__global__ __launch_bounds__(BLOCK) void successor_kernel(u64 N, const u64* __restrict__ first_out, const u64* __restrict__ dst,
                                                          u64* __restrict__ node_deg) {
    for (u64 v = (u64)blockIdx.x * BLOCK + threadIdx.x; v < N; v += (u64)gridDim.x * BLOCK)
        node_deg[v] = with_successor(node_deg[v], first_out[v], dst);
}

// check_dead_path(vertex, Incoming, Outgoing) (pruner.rs:229-257): returns the number of edges of the dead path, 0 if
// the walk is not dead; with MARK, adds one to mult[] of every edge on the way
// Per vertex, the positions of its out-edges by the edge's LAST base and of its in-edges by the edge's FIRST base: a
// (k-1)-mer has at most four of each and they differ in exactly that base, so a slot is found without searching and kept
// current in O(1) when an edge moves.  With them a pass touches only what it changes: the first out-edge of a vertex that
// lost it is the youngest of its (at most three) remaining out-edges, and the edges whose endpoint was
// re-numbered are the (at most eight) edges in the moved vertex's slots -- no pass over all edges.
struct Slots { u32* out; u32* in; u32 nw, k; };
__device__ __forceinline__ u32 last_base(const u64* __restrict__ key, u64 e, u32 nw) { return (u32)(key[e * nw + nw - 1] & 3); }
__device__ __forceinline__ u32 first_base(const u64* __restrict__ key, u64 e, u32 nw, u32 k) {
    const u32 bit = 2 * (k - 1);                                       // of the right-aligned k-mer, w[0] the high word
    return (u32)(key[e * nw + (nw - 1 - bit / 64)] >> (bit % 64)) & 3;
}
__global__ __launch_bounds__(BLOCK) void slots_init_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, const u64* __restrict__ key,
                                                           u64 E, Slots sl) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        sl.out[src[e] * 4 + last_base(key, e, sl.nw)] = (u32)e;
        sl.in[dst[e] * 4 + first_base(key, e, sl.nw, sl.k)] = (u32)e;
    }
}
// vertices that lost their first out-edge in this pass (listed by death_count_kernel): the youngest out-edge left
__global__ __launch_bounds__(BLOCK) void redo_slots_kernel(const u32* __restrict__ list, u64 n, Slots sl, const u32* __restrict__ orig,
                                                           const u64* __restrict__ dst, u64* __restrict__ node_deg, u64* __restrict__ first_out) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 a = list[i];
        u64 best = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const u32 e = sl.out[a * 4 + b];
            if (e != NONE32) { const u64 c = ((u64)(orig[e] + 1u) << 32) | e; best = c > best ? c : best; }
        }
        first_out[a] = best;
        node_deg[a] = with_successor(node_deg[a], best, dst);
    }
}
// vertices that were moved (to[i] <- from[i]): their edges name them by the new id; a vertex whose first out-edge leads to
// a moved vertex gets the new successor
__global__ __launch_bounds__(BLOCK) void remap_slots_kernel(const u32* __restrict__ to, u64 n, u64 n_new, const u32* __restrict__ tail_map, Slots sl,
                                                            u64* __restrict__ src, u64* __restrict__ dst, const u64* __restrict__ first_out,
                                                            u64* __restrict__ node_deg) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 d = to[i];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const u32 eo = sl.out[d * 4 + b];
            if (eo != NONE32) src[eo] = d;
            const u32 ei = sl.in[d * 4 + b];
            if (ei != NONE32) {
                dst[ei] = d;
                u64 p = src[ei];                                      // (its own move may or may not have been written yet)
                if (p >= n_new) p = tail_map[p - n_new];
                if ((u32)first_out[p] == ei && first_out[p] != 0) node_deg[p] = (node_deg[p] & 0xFFFFFFFFull) | (d << 32);
            }
        }
    }
}

__device__ __forceinline__ u32 walk_length(u32 v, u32 two_k, const u64* __restrict__ node_deg) {
    u64 w = node_deg[v];
    u32 cnt = 0, n = 0;
    for (;;) {
        cnt += 1;
        if (cnt >= two_k) return 0;                             // "this path is not dead"
        const u32 next = (u32)(w >> 32);
        if (next == NONE32) return n;                           // no out-edge: the whole path is dead
        ++n;
        w = node_deg[next];
        if ((w & IN_MASK) >= 3) return n;                       // neighbors_directed(current, Incoming).nth(2).is_some()
    }
}
// the same walk over the edges themselves: one more listing of each of its `len` edges
constexpr u32 MARK_GROUP_SHIFT = 11;       // 2048 edges (one workgroup of the mark compaction) share a "something is marked" byte
__device__ __forceinline__ void walk_mark(u32 v, u32 len, const u64* __restrict__ first_out, const u64* __restrict__ dst, u32* __restrict__ mult,
                                          unsigned char* __restrict__ touched) {
    u32 cur = v;
    for (u32 n = 0; n < len; ++n) {
        const u32 e = (u32)first_out[cur];
        atomicAdd(&mult[e], 1u);
        touched[e >> MARK_GROUP_SHIFT] = 1;
        cur = (u32)dst[e];
    }
}

__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Externals (pruner.rs:165-195): Input = no incoming edge.  (A vertex with no edge at all cannot exist here.)  The
// Input vertices are gathered into a list first -- they are a few per cent of the vertices, and a walk is up to 2k
// dependent look-ups long: walking from inside the scan leaves most lanes of a workgroup idle for that long.
constexpr u32 INPUT_BUF = 2048;        // Input vertices a workgroup collects in LDS before it claims room in the list
__global__ __launch_bounds__(BLOCK) void input_list_kernel(u64 N, u64* __restrict__ node_deg, const u64* __restrict__ first_out,
                                                           const u64* __restrict__ dst, const u32* __restrict__ tail_map /* of the pass before */,
                                                           u32* __restrict__ list, u64* __restrict__ totals /* [2] walks */, bool keep_list) {
    __shared__ u32 buf[INPUT_BUF];
    __shared__ u32 cnt;
    __shared__ u64 base;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const u32 lane = threadIdx.x & 63;
    const u64 step = (u64)gridDim.x * BLOCK;
    for (u64 v0 = (u64)blockIdx.x * BLOCK; v0 < N; v0 += step) {          // the whole workgroup stays in the loop together
        const u64 v = v0 + threadIdx.x;
        bool input = false;
        if (v < N) {
            const u64 deg = node_deg[v];
            const u32 next = (u32)(deg >> 32);
            if (deg & REDO_FIRST_OUT) node_deg[v] = with_successor(deg & ~REDO_FIRST_OUT, first_out[v], dst);   // a new first out-edge
            else if (tail_map && next != NONE32 && next >= N) node_deg[v] = (deg & DEG_MASK) | ((u64)tail_map[next - N] << 32);   // it was moved
            input = (deg & IN_MASK) == 0;
            if (input && keep_list && !(deg & LISTED)) node_deg[v] = (deg & ~REDO_FIRST_OUT) | LISTED;   // (no first out-edge is lost before the first pass)
        }
        const u64 m = __ballot(input);
        if (m) {
            u32 at = 0;
            if (lane == 0) at = atomicAdd(&cnt, (u32)__popcll(m));
            at = __shfl(at, 0, 64);
            if (input) buf[at + __popcll(m & (lane ? (~0ull >> (64 - lane)) : 0ull))] = (u32)v;
        }
        __syncthreads();
        const u32 have = cnt;
        __syncthreads();                                                   // (everybody has looked before thread 0 may reset the count)
        const bool last = v0 + step >= N;
        if (have > INPUT_BUF - BLOCK || (last && have)) {                  // one claim on the shared counter per ~2000 vertices
            if (threadIdx.x == 0) { base = atomicAdd((unsigned long long*)&totals[2], (unsigned long long)have); cnt = 0; }
            __syncthreads();
            for (u32 j = threadIdx.x; j < have; j += BLOCK) list[base + j] = buf[j];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void walk_kernel(const u32* __restrict__ list, u32 two_k, const u64* __restrict__ first_out,
                                                     const u64* __restrict__ dst, const u64* __restrict__ node_deg, u32* __restrict__ mult,
                                                     unsigned char* __restrict__ touched,
                                                     const u32* __restrict__ stamp, u32 walk_from /* 0: every listed vertex */, bool check,
                                                     u64* __restrict__ totals /* [0] marks, [1] dead walks, [2] listed (input), [3] walked,
                                                                                 [4] (check) dead walks that would have been skipped */) {
    const u64 n = totals[2];
    u32 marks = 0, dead = 0, walked = 0;
    for (u64 j = (u64)blockIdx.x * BLOCK + threadIdx.x; j < n; j += (u64)gridDim.x * BLOCK) {
        const u32 v = list[j];
        const bool skip = walk_from && stamp[v] < walk_from;   // nothing within reach of its walk has changed: still not dead
        if (skip && !check) continue;
        walked += 1;
        const u32 len = walk_length(v, two_k, node_deg);
        if (len) {
            walk_mark(v, len, first_out, dst, mult, touched); marks += len; dead += 1;
            if (skip) atomicAdd((unsigned long long*)&totals[4], 1ull);          // KATOME_PRUNE_CHECK_WALKS: must never happen
        }
    }
    marks = wave_sum(marks); dead = wave_sum(dead); walked = wave_sum(walked);
    if ((threadIdx.x & 63) == 0) {
        if (marks) atomicAdd((unsigned long long*)&totals[0], (unsigned long long)marks);
        if (dead) atomicAdd((unsigned long long*)&totals[1], (unsigned long long)dead);
        if (walked) atomicAdd((unsigned long long*)&totals[3], (unsigned long long)walked);
    }
}

// marked edge indices, ascending, with their multiplicity: count / scan / write
__global__ __launch_bounds__(BLOCK) void mark_count_kernel(const u32* __restrict__ mult, u64 E, u32* __restrict__ counts,
                                                           const unsigned char* __restrict__ touched = nullptr) {
    __shared__ u32 total;
    if (touched && !touched[blockIdx.x]) {                 // nothing marked among this workgroup's edges: mult is not read
        if (threadIdx.x == 0) counts[blockIdx.x] = 0;
        return;
    }
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const u64 base = ((u64)blockIdx.x * BLOCK + threadIdx.x) * MARK_ITEMS;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j) c += (base + j < E && mult[base + j] != 0);
    if (c) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(BLOCK) void mark_write_kernel(const u32* __restrict__ mult, u64 E, const u64* __restrict__ block_offs,
                                                           u32* __restrict__ out_pos, u32* __restrict__ out_mult,
                                                           const unsigned char* __restrict__ touched) {
    __shared__ u32 wsum[BLOCK / 64];
    if (!touched[blockIdx.x]) return;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = ((u64)blockIdx.x * BLOCK + tid) * MARK_ITEMS;
    u32 m[MARK_ITEMS], c = 0;
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j) { m[j] = base + j < E ? mult[base + j] : 0; c += m[j] != 0; }
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < wave; ++w) woff += wsum[w];
    u64 pos = block_offs[blockIdx.x] + woff + incl - c;
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j) if (m[j]) { out_pos[pos] = (u32)(base + j); out_mult[pos] = m[j]; ++pos; }
}

__global__ __launch_bounds__(BLOCK) void mark_clear_kernel(const u32* __restrict__ pos, u64 n, u32* __restrict__ mult,
                                                           unsigned char* __restrict__ touched) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) { mult[pos[i]] = 0; touched[pos[i] >> MARK_GROUP_SHIFT] = 0; }
}

// which endpoints lose their last edge with removal t: take the removed edges out of the degree words, remember the
// last removal that touched each node, then ask per removal
__global__ __launch_bounds__(BLOCK) void death_count_kernel(const u32* __restrict__ victims, u64 m, const u64* __restrict__ src,
                                                            const u64* __restrict__ dst, u64* __restrict__ node_deg,
                                                            u64* __restrict__ first_out, u32* __restrict__ last_touch,
                                                            Slots sl, const u64* __restrict__ key, u32* __restrict__ redo_list,
                                                            u64* __restrict__ redo_count) {
    const u32 lane = threadIdx.x & 63;
    const u64 step = (u64)gridDim.x * BLOCK;
    for (u64 t0 = (u64)blockIdx.x * BLOCK; t0 < m; t0 += step) {            // whole waves stay in the loop together
        const u64 t = t0 + threadIdx.x;
        bool head = false;
        u64 a = 0;
        if (t < m) {
            const u32 e = victims[t];
            a = src[e];
            const u64 b = dst[e];
            const u64 fo = first_out[a];
            head = fo != 0 && (u32)fo == e;                                  // the list head goes
            if (head) {
                first_out[a] = 0;
                if (!sl.out) atomicOr((unsigned long long*)&node_deg[a], REDO_FIRST_OUT);   // found again by first_out_redo_kernel
            }
            if (sl.out) { sl.out[a * 4 + last_base(key, e, sl.nw)] = NONE32; sl.in[b * 4 + first_base(key, e, sl.nw, sl.k)] = NONE32; }
            atomicAdd((unsigned long long*)&node_deg[a], 0ull - OUT_ONE);
            atomicAdd((unsigned long long*)&node_deg[b], 0ull - IN_ONE);
            atomicMax(&last_touch[a], (u32)t + 1u);
            atomicMax(&last_touch[b], (u32)t + 1u);
        }
        if (sl.out) {                                                        // ... or listed for redo_slots_kernel
            const u64 mask = __ballot(head);
            if (mask) {
                u64 base = 0;
                const int leader = __ffsll((unsigned long long)mask) - 1;
                if ((int)lane == leader) base = atomicAdd((unsigned long long*)redo_count, (unsigned long long)__popcll(mask));
                base = __shfl(base, leader, 64);
                if (head) redo_list[base + __popcll(mask & (lane ? (~0ull >> (64 - lane)) : 0ull))] = (u32)a;
            }
        }
    }
}
// The Input vertices of the next pass without looking at every vertex: those of this pass that are still there (a vertex
// with no in-edge cannot gain one) plus the vertices whose last in-edge this pass removed, minus the ones that died, under
// their new ids where they were moved.
__global__ __launch_bounds__(BLOCK) void inputs_update_kernel(const u32* __restrict__ old_list, u64 n_old, const u32* __restrict__ fresh, u64 n_fresh,
                                                              const u32* __restrict__ last_touch, u64 n_new, const u32* __restrict__ tail_map,
                                                              u64* __restrict__ node_deg, u32* __restrict__ out, u64* __restrict__ count,
                                                              u32* __restrict__ stamp, u32 walk_code) {
    __shared__ u32 wcnt[BLOCK / 64];
    __shared__ u64 bbase;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 total = n_old + n_fresh, step = (u64)gridDim.x * BLOCK;
    for (u64 i0 = (u64)blockIdx.x * BLOCK; i0 < total; i0 += step) {
        const u64 i = i0 + tid;
        bool keep = false;
        u32 v = 0;
        if (i < total) {
            v = i < n_old ? old_list[i] : fresh[i - n_old];
            keep = last_touch[v] != DIED;                                  // (ids of the pass that just ended)
            if (keep && v >= n_new) v = tail_map[v - n_new];
            if (keep && i >= n_old) { node_deg[v] |= LISTED; if (stamp) atomicMax(&stamp[v], walk_code); }   // (one entry per vertex: no other writer)
        }
        const u64 mask = __ballot(keep);
        if (lane == 0) wcnt[wave] = (u32)__popcll(mask);
        __syncthreads();
        if (tid == 0) {
            u32 tot = 0;
            for (int w = 0; w < BLOCK / 64; ++w) tot += wcnt[w];
            bbase = tot ? atomicAdd((unsigned long long*)count, (unsigned long long)tot) : 0;
        }
        __syncthreads();
        if (keep) {
            u32 woff = 0;
            for (u32 w = 0; w < wave; ++w) woff += wcnt[w];
            out[bbase + woff + __popcll(mask & (lane ? (~0ull >> (64 - lane)) : 0ull))] = v;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(BLOCK) void death_emit_kernel(const u32* __restrict__ victims, u64 m, const u64* __restrict__ src,
                                                           const u64* __restrict__ dst, const u64* __restrict__ node_deg,
                                                           u32* last_touch, u32* __restrict__ die,
                                                           u32* __restrict__ fresh /* nullptr: no list of Input vertices is kept */, u64* __restrict__ fresh_count,
                                                           u32* __restrict__ changed /* nullptr: not wanted */, u64* __restrict__ changed_count) {
    const u32 lane = threadIdx.x & 63;
    const u64 step = (u64)gridDim.x * BLOCK;
    for (u64 t0 = (u64)blockIdx.x * BLOCK; t0 < m; t0 += step) {            // whole waves stay in the loop together
        const u64 t = t0 + threadIdx.x;
        u32 f0 = NONE32, f1 = NONE32;                                        // endpoints that have just become Input vertices
        u32 c0 = NONE32, c1 = NONE32;                                        // endpoints that stay, with a degree or a successor changed
        if (t < m) {
            const u32 e = victims[t];
            const u64 a = src[e], b = dst[e];
            const u64 wa = node_deg[a], wb = node_deg[b];
            // the removal that touched a vertex last speaks for it (exactly one does)
            const bool mine_a = last_touch[a] == (u32)t + 1u, mine_b = b != a && last_touch[b] == (u32)t + 1u;
            const bool da = mine_a && (wa & DEGREES) == 0, db = mine_b && (wb & DEGREES) == 0;
            die[2 * t] = da ? (u32)a : NONE32;
            die[2 * t + 1] = db ? (u32)b : NONE32;
            if (fresh) {
                // (the other removals that touched a dying vertex compare its last_touch with their own number: neither the
                // old value nor DIED matches, so this store may land while they look)
                if (da) last_touch[a] = DIED;
                if (db) last_touch[b] = DIED;
                if (mine_a && !da && (wa & IN_MASK) == 0 && !(wa & LISTED)) f0 = (u32)a;
                if (mine_b && !db && (wb & IN_MASK) == 0 && !(wb & LISTED)) f1 = (u32)b;
                if (changed) { if (mine_a && !da) c0 = (u32)a; if (mine_b && !db) c1 = (u32)b; }
            }
        }
        if (changed) {
            const u64 m0 = __ballot(c0 != NONE32), m1 = __ballot(c1 != NONE32);
            if (m0 | m1) {
                u64 base = 0;
                if (lane == 0) base = atomicAdd((unsigned long long*)changed_count, (unsigned long long)(__popcll(m0) + __popcll(m1)));
                base = __shfl(base, 0, 64);
                const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
                if (c0 != NONE32) changed[base + __popcll(m0 & below)] = c0;
                if (c1 != NONE32) changed[base + __popcll(m0) + __popcll(m1 & below)] = c1;
            }
        }
        if (fresh) {
            const u64 m0 = __ballot(f0 != NONE32), m1 = __ballot(f1 != NONE32);
            if (m0 | m1) {
                u64 base = 0;
                if (lane == 0) base = atomicAdd((unsigned long long*)fresh_count, (unsigned long long)(__popcll(m0) + __popcll(m1)));
                base = __shfl(base, 0, 64);
                const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
                if (f0 != NONE32) fresh[base + __popcll(m0 & below)] = f0;
                if (f1 != NONE32) fresh[base + __popcll(m0) + __popcll(m1 & below)] = f1;
            }
        }
    }
}
// Which listed Input vertices must be walked again: a walk reads, for up to 2k vertices along first out-edges, each
// vertex's successor and in-degree -- so its outcome can only differ from the pass before if one of the vertices whose
// degree or successor this pass changed lies within 2k steps of it.  From every such vertex the vertices that reach it
// along their first out-edges are visited backwards (in-edge slots, kept only where the source's first out-edge is that
// edge), depth first with the smallest depth seen per vertex stamped (pass number and depth in one word, atomicMax), and
// every vertex visited is stamped "walk in this pass"; the walks then skip the listed vertices without the stamp.
// stamp = pass number * STAMP_STEP + (STAMP_STEP - 1 - depth): a later pass, or the same pass nearer to a change, is larger
constexpr u32 STAMP_STEP = 256;            // depths go to 2k <= 126
constexpr int WALK_STACK = 3 * 126 + 8;
__global__ __launch_bounds__(BLOCK) void affected_kernel(const u32* __restrict__ changed, u64 n, u64 n_new, const u32* __restrict__ tail_map,
                                                         Slots sl, const u64* __restrict__ src, const u64* __restrict__ node_deg,
                                                         u32* __restrict__ stamp, u32 code0 /* this pass, depth 0 */, u32 max_depth) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        u32 x = changed[i];
        if (x >= n_new) x = tail_map[x - n_new];
        u32 sv[WALK_STACK]; unsigned char sd[WALK_STACK];
        int top = 0;
        sv[0] = x; sd[0] = 0; top = 1;
        while (top) {
            --top;
            const u32 v = sv[top]; const u32 d = sd[top];
            const u32 code = code0 - d;                               // shallower = larger
            if (atomicMax(&stamp[v], code) >= code) continue;          // seen in this pass at this depth or nearer
            if (d == max_depth) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const u32 e = sl.in[(u64)v * 4 + b];
                if (e == NONE32) continue;
                const u32 p = (u32)src[e];
                if ((u32)(node_deg[p] >> 32) != v) continue;          // p's walk does not come this way
                if (top == WALK_STACK) continue;                      // (cannot happen: 3 * depth + 4 entries at most)
                sv[top] = p; sd[top] = (unsigned char)(d + 1); ++top;
            }
        }
    }
}

// moves: every array entry of the edge (node) at `from` goes to `to`; sources lie at or above the new count and
// targets below it, so the copies never overlap
__global__ __launch_bounds__(BLOCK) void move_edges_kernel(const u32* __restrict__ to, const u32* __restrict__ from, u64 n, u32 nw,
                                                           u64* __restrict__ src, u64* __restrict__ dst, u32* __restrict__ weight,
                                                           u32* __restrict__ orig, u64* __restrict__ key, u64* __restrict__ first_out,
                                                           Slots sl) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 d = to[i], s = from[i];
        const u64 a = src[s];
        if (sl.out) { sl.out[a * 4 + last_base(key, s, sl.nw)] = (u32)d; sl.in[dst[s] * 4 + first_base(key, s, sl.nw, sl.k)] = (u32)d; }
        if (first_out) {
            const u64 fo = first_out[a];
            if (fo != 0 && (u32)fo == (u32)s) first_out[a] = (fo & 0xFFFFFFFF00000000ull) | d;  // the head follows its edge
        }
        src[d] = a; dst[d] = dst[s]; weight[d] = weight[s]; orig[d] = orig[s];
        for (u32 w = 0; w < nw; ++w) key[d * nw + w] = key[s * nw + w];
    }
}
__global__ __launch_bounds__(BLOCK) void move_nodes_kernel(const u32* __restrict__ to, const u32* __restrict__ from, u64 n, u32 nw,
                                                           u64 n_new, u64* __restrict__ node_key, u64* __restrict__ node_deg,
                                                           u64* __restrict__ first_out, u32* __restrict__ tail_map,
                                                           Slots sl) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) {
        const u64 d = to[i], s = from[i];
        for (u32 w = 0; w < nw; ++w) node_key[d * nw + w] = node_key[s * nw + w];
        if (node_deg) { node_deg[d] = node_deg[s]; first_out[d] = first_out[s]; }
        if (sl.out) {
            reinterpret_cast<uint4*>(sl.out)[d] = reinterpret_cast<const uint4*>(sl.out)[s];
            reinterpret_cast<uint4*>(sl.in)[d] = reinterpret_cast<const uint4*>(sl.in)[s];
        }
        tail_map[s - n_new] = (u32)d;
    }
}
__global__ __launch_bounds__(BLOCK) void remap_kernel(u64* __restrict__ src, u64* __restrict__ dst, u64 E, u64 n_new,
                                                      const u32* __restrict__ tail_map) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        const u64 a = src[e], b = dst[e];
        if (a >= n_new) src[e] = tail_map[a - n_new];
        if (b >= n_new) dst[e] = tail_map[b - n_new];
    }
}

// Clean::remove_weak_edges (pruner.rs:84-93): flags for the two retain passes
__global__ __launch_bounds__(BLOCK) void weak_flag_kernel(const u32* __restrict__ weight, u64 E, u32 threshold, u32* __restrict__ flag) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) flag[e] = weight[e] < threshold ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void touch_nodes_kernel(const u64* __restrict__ src, const u64* __restrict__ dst, u64 E,
                                                            u32* __restrict__ lonely) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) { lonely[src[e]] = 0; lonely[dst[e]] = 0; }
}
__global__ __launch_bounds__(BLOCK) void fill_u32_kernel(u32* __restrict__ p, u64 n, u32 v) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) p[i] = v;
}

// one pass over the edges that are left: endpoints that were moved get their new ids (REMAP), and nodes that lost
// their first out-edge find the largest first-seen index among the out-edges they still have
template <bool REMAP>
__global__ __launch_bounds__(BLOCK) void first_out_redo_kernel(u64* __restrict__ src, u64* __restrict__ dst, const u32* __restrict__ orig, u64 E,
                                                               u64 n_new, const u32* __restrict__ tail_map,
                                                               const u64* __restrict__ node_deg, u64* __restrict__ first_out) {
    for (u64 e = (u64)blockIdx.x * BLOCK + threadIdx.x; e < E; e += (u64)gridDim.x * BLOCK) {
        u64 a = src[e];
        if (REMAP) {
            if (a >= n_new) { a = tail_map[a - n_new]; src[e] = a; }
            const u64 b = dst[e];
            if (b >= n_new) dst[e] = tail_map[b - n_new];
        }
        if (node_deg[a] & REDO_FIRST_OUT)
            atomicMax((unsigned long long*)&first_out[a], ((unsigned long long)(orig[e] + 1u) << 32) | (unsigned long long)e);
    }
}

// grow-only pinned host buffer: the per-pass device -> host copies run at link speed and stay asynchronous
struct PinnedU32 {
    u32* p = nullptr; size_t cap = 0;
    ~PinnedU32() { if (p) (void)hipHostFree(p); }
    int need(size_t n) {
        if (n <= cap) return KATOME_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 1024;
        if (hipHostMalloc((void**)&p, want * 4, hipHostMallocDefault) != hipSuccess) { p = nullptr; set_error("out of pinned host memory (%zu bytes)", want * 4); return KATOME_E_OOM; }
        cap = want;
        return KATOME_OK;
    }
};

// ---- remove_edge replay on the device ---------------------------------------------------------------------------------
// The marked positions are consumed from the top (prune_replay.h, replay_edges, is the sequential statement).  With n the
// number of edges left, an entry (position d, listed c times) takes min(c, n - d) edges away -- the one at d, then
// whatever was moved into d -- and leaves n' = max(n - c, d): a function n -> max(n - a, b), and such functions compose
// to functions of the same shape, (a1, b1) then (a2, b2) = (a1 + a2, max(b1 - a2, b2)).  A scan over the entries, top
// first, therefore gives every entry the count it starts from, and with it the ordinal of its first removal and the
// positions its movers come from (n - 1, n - 2, ...).  What sits at such a position is the edge that started there, or,
// if the position is itself a marked one, whatever that entry left in it: chains that only run upwards, followed by
// pointer jumping over the tail that disappears (as retain_on_device does).
struct Clamp { long long a, b; };
__device__ __forceinline__ Clamp clamp_then(const Clamp& f, const Clamp& g) {      // f first, then g
    Clamp r; r.a = f.a + g.a; r.b = (f.b - g.a) > g.b ? (f.b - g.a) : g.b; return r;
}
__device__ __forceinline__ long long clamp_apply(const Clamp& f, long long n) { return (n - f.a) > f.b ? (n - f.a) : f.b; }
constexpr int REPLAY_ITEMS = 8;
constexpr long long CLAMP_NONE = -(1ll << 60);

// entries are numbered r = 0.. from the top: r <-> ascending index u - 1 - r.  MODE 0: the workgroup's composed function
// -> agg[block].  MODE 1: carry[block] is what precedes the workgroup; size_before[i] for every entry of it, and the
// count left after the very last entry -> totals[3].
// (I: the index type -- u32 on one GPU, u64 for the positions of a graph sharded over several, dist_prune.hip)
template <int MODE, class I>
__global__ __launch_bounds__(BLOCK) void replay_scan_kernel(const I* __restrict__ pos, const u32* __restrict__ mult, u64 u, u64 n_edges,
                                                            Clamp* __restrict__ agg, const Clamp* __restrict__ carry,
                                                            I* __restrict__ size_before, u64* __restrict__ totals) {
    __shared__ Clamp part[BLOCK];
    const u32 tid = threadIdx.x;
    const u64 r0 = ((u64)blockIdx.x * BLOCK + tid) * REPLAY_ITEMS;
    Clamp f; f.a = 0; f.b = CLAMP_NONE;
#pragma unroll
    for (int j = 0; j < REPLAY_ITEMS; ++j) {
        const u64 r = r0 + j;
        if (r < u) { Clamp g; g.a = mult[u - 1 - r]; g.b = pos[u - 1 - r]; f = clamp_then(f, g); }
    }
    part[tid] = f;
    __syncthreads();
    for (u32 o = 1; o < BLOCK; o <<= 1) {                    // inclusive scan of the threads' functions (order matters)
        Clamp mine = part[tid], left;
        const bool take = tid >= o;
        if (take) left = part[tid - o];
        __syncthreads();
        if (take) part[tid] = clamp_then(left, mine);
        __syncthreads();
    }
    if (MODE == 0) {
        if (tid == BLOCK - 1) agg[blockIdx.x] = part[tid];
        return;
    }
    Clamp before = carry[blockIdx.x];
    if (tid) before = clamp_then(before, part[tid - 1]);
    long long n = clamp_apply(before, (long long)n_edges);
#pragma unroll
    for (int j = 0; j < REPLAY_ITEMS; ++j) {
        const u64 r = r0 + j;
        if (r < u) {
            const u64 i = u - 1 - r;
            size_before[i] = (I)n;
            const long long left = n - (long long)mult[i], d = pos[i];
            n = left > d ? left : d;
            if (i == 0) totals[3] = (u64)n;
        }
    }
}
// one workgroup: carry[b] = the functions of the workgroups before b, composed
__global__ __launch_bounds__(BLOCK) void replay_spine_kernel(const Clamp* __restrict__ agg, u64 nb, Clamp* __restrict__ carry) {
    __shared__ Clamp part[BLOCK];
    const u32 tid = threadIdx.x;
    const u64 per = (nb + BLOCK - 1) / BLOCK, b0 = (u64)tid * per, b1 = b0 + per < nb ? b0 + per : nb;
    Clamp f; f.a = 0; f.b = CLAMP_NONE;
    for (u64 b = b0; b < b1; ++b) f = clamp_then(f, agg[b]);
    part[tid] = f;
    __syncthreads();
    for (u32 o = 1; o < BLOCK; o <<= 1) {
        Clamp mine = part[tid], left;
        const bool take = tid >= o;
        if (take) left = part[tid - o];
        __syncthreads();
        if (take) part[tid] = clamp_then(left, mine);
        __syncthreads();
    }
    Clamp run; run.a = 0; run.b = CLAMP_NONE;
    if (tid) run = part[tid - 1];
    for (u64 b = b0; b < b1; ++b) { carry[b] = run; run = clamp_then(run, agg[b]); }
}
template <class I>
__global__ __launch_bounds__(BLOCK) void iota_from_kernel(I* __restrict__ out, u64 n, I first) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * BLOCK) out[i] = first + (I)i;
}
// marked positions inside the tail [M, n): where their last occupant comes from; totals[4] = entries below M
template <class I>
__global__ __launch_bounds__(BLOCK) void replay_links_kernel(const I* __restrict__ pos, const u32* __restrict__ mult,
                                                             const I* __restrict__ size_before, u64 u, u64 M, I* __restrict__ jump,
                                                             u64* __restrict__ totals) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < u; i += (u64)gridDim.x * BLOCK) {
        const u64 d = pos[i];
        if (d < M) continue;
        if (i == 0 || pos[i - 1] < M) totals[4] = i;
        const long long left = (long long)size_before[i] - (long long)mult[i];
        if (left > (long long)d) jump[d - M] = (I)left;          // (else the entry ends by taking the last edge itself)
    }
}
// victims in removal order, the moves into the marked positions that stay, the removals owed to repeated indices
template <class I>
__global__ __launch_bounds__(BLOCK) void replay_emit_kernel(const I* __restrict__ pos, const u32* __restrict__ mult,
                                                            const I* __restrict__ size_before, u64 u, u64 n_edges, u64 M,
                                                            const I* __restrict__ jump, I* __restrict__ victims,
                                                            I* __restrict__ move_to, I* __restrict__ move_from, u64* __restrict__ totals) {
    u32 dups = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < u; i += (u64)gridDim.x * BLOCK) {
        const u64 d = pos[i], S = size_before[i];
        const long long left = (long long)S - (long long)mult[i];
        const u64 after = left > (long long)d ? (u64)left : d, e = S - after, first = n_edges - S;
        victims[first] = (I)d;
        for (u64 j = 1; j < e; ++j) victims[first + j] = jump[S - j - M];
        dups += (u32)(e - 1);
        if (d < M) { move_to[i] = (I)d; move_from[i] = jump[after - M]; }
    }
    dups = wave_sum(dups);
    if ((threadIdx.x & 63) == 0 && dups) atomicAdd((unsigned long long*)&totals[5], (unsigned long long)dups);
}

// ---- remove_node replay on the device ---------------------------------------------------------------------------------
// replay_nodes (prune_replay.h) is the sequential statement.  The M nodes that die leave one at a time; with removal
// number s the last position L(s) = N - 1 - s is given up and whoever sits there moves into the position p(s) of the
// node being removed.  Only positions of the tail [N - M, N) are ever given up, each exactly once and in descending
// order, so "where does the occupant of tail position q go" is one number per tail position: hole[q] = p(N - 1 - q).
// A node that started at x is, at removal s, at the end of the chain x -> hole[x] -> hole[hole[x]] ... cut where the
// next position has not been given up yet (N - 1 - q >= s) or lies below the tail.  p(s) itself is such a chain end (for
// the node removed at s, over positions given up before s): the holes depend on holes of higher positions only.  Most
// chains have one or two links, so the holes are found in a few rounds: every removal whose chain runs over known
// holes settles, the rest wait for the next round.  When an edge removal takes both endpoints, the endpoint at the
// larger current position goes first (pruner.rs:206-225) -- both positions are chain ends at the same removal number.
template <class I> struct IdxNone { static constexpr I value = (I)~(I)0; };      // REPLAY_NONE / "hole not known yet" in either index width
constexpr u32 CHAIN_CAP = 1u << 13;        // links followed per chain before the pass is handed to the host replay

template <class I>
__global__ __launch_bounds__(BLOCK) void die_count_kernel(const I* __restrict__ die, u64 m, u32* __restrict__ counts) {
    constexpr I NONE32 = IdxNone<I>::value;
    __shared__ u32 total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const u64 base = ((u64)blockIdx.x * BLOCK + threadIdx.x) * MARK_ITEMS;
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j)
        if (base + j < m) c += (die[2 * (base + j)] != NONE32) + (die[2 * (base + j) + 1] != NONE32);
    if (c) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
// first[t] = nodes removed before edge removal t; dead[] marks the tail nodes that die
template <class I>
__global__ __launch_bounds__(BLOCK) void die_first_kernel(const I* __restrict__ die, u64 m, const u64* __restrict__ block_offs, u64 base_pos,
                                                          I* __restrict__ first, unsigned char* __restrict__ dead) {
    constexpr I NONE32 = IdxNone<I>::value;
    __shared__ u32 wsum[BLOCK / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = ((u64)blockIdx.x * BLOCK + tid) * MARK_ITEMS;
    u32 cnt[MARK_ITEMS], c = 0;
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j) {
        cnt[j] = 0;
        if (base + j < m) {
            const I a = die[2 * (base + j)], b = die[2 * (base + j) + 1];
            cnt[j] = (a != NONE32) + (b != NONE32);
            if (a != NONE32 && a >= base_pos) dead[a - base_pos] = 1;
            if (b != NONE32 && b >= base_pos) dead[b - base_pos] = 1;
        }
        c += cnt[j];
    }
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < wave; ++w) woff += wsum[w];
    u64 run = block_offs[blockIdx.x] + woff + incl - c;
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j) if (base + j < m) { first[base + j] = (I)run; run += cnt[j]; }
}
// position at removal number s of the node that started at x (alive then); false: a hole on the way is not known yet
template <class I>
__device__ __forceinline__ bool node_position(I x, u64 s, u64 N, u64 base_pos, const I* hole, u32* flags, I& out) {
    constexpr I UNRESOLVED = IdxNone<I>::value;
    u64 q = x;
    u32 links = 0;
    while (q >= base_pos && N - 1 - q < s) {
        const I h = hole[q - base_pos];
        if (h == UNRESOLVED) return false;
        if (h == q) break;                                   // (the occupant died here; no live node follows this link)
        q = h;
        if (++links > CHAIN_CAP) { flags[1] = 1; return false; }
    }
    out = (I)q;
    return true;
}
template <class I>
__global__ __launch_bounds__(BLOCK) void node_holes_kernel(const I* __restrict__ die, const I* __restrict__ first, u64 m, u64 N,
                                                           u64 base_pos, I* hole, u32* flags /* [0] something waits, [1] chain too long */) {
    constexpr I NONE32 = IdxNone<I>::value, UNRESOLVED = IdxNone<I>::value;
    bool waits = false;
    for (u64 t = (u64)blockIdx.x * BLOCK + threadIdx.x; t < m; t += (u64)gridDim.x * BLOCK) {
        const I a = die[2 * t], b = die[2 * t + 1];
        if (a == NONE32 && b == NONE32) continue;
        const u64 s = first[t], last = N - 1 - s;
        if (hole[last - base_pos] != UNRESOLVED) continue;                   // settled in an earlier round
        if (a != NONE32 && b != NONE32) {
            I pa, pb;
            if (!node_position(a, s, N, base_pos, hole, flags, pa) || !node_position(b, s, N, base_pos, hole, flags, pb)) { waits = true; continue; }
            hole[last - 1 - base_pos] = pa < pb ? pa : pb;                      // the second to go (it cannot be the one sitting last)
            hole[last - base_pos] = pa < pb ? pb : pa;
        } else {
            I p;
            if (!node_position(a != NONE32 ? a : b, s, N, base_pos, hole, flags, p)) { waits = true; continue; }
            hole[last - base_pos] = p;
        }
    }
    if (waits) flags[0] = 1;
}
// the tail nodes that stay: where each ends up
template <class I>
__global__ __launch_bounds__(BLOCK) void node_moves_kernel(const unsigned char* __restrict__ dead, u64 M, u64 base_pos, const I* __restrict__ hole,
                                                           I* __restrict__ move_to, I* __restrict__ move_from, u64* __restrict__ count,
                                                           u32* flags) {
    constexpr I UNRESOLVED = IdxNone<I>::value;
    __shared__ u32 wcnt[BLOCK / 64];
    __shared__ u64 bbase;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 step = (u64)gridDim.x * BLOCK;
    for (u64 i0 = (u64)blockIdx.x * BLOCK; i0 < M; i0 += step) {
        const u64 i = i0 + tid;
        bool stays = i < M && !dead[i];
        u64 q = base_pos + i;
        if (stays) {
            u32 links = 0;
            while (q >= base_pos) {
                const I h = hole[q - base_pos];
                if (h == UNRESOLVED || h == q || ++links > CHAIN_CAP) { flags[1] = 1; stays = false; break; }
                q = h;
            }
        }
        const u64 mask = __ballot(stays);
        if (lane == 0) wcnt[wave] = (u32)__popcll(mask);
        __syncthreads();
        if (tid == 0) {
            u32 tot = 0;
            for (int w = 0; w < BLOCK / 64; ++w) tot += wcnt[w];
            bbase = tot ? atomicAdd((unsigned long long*)count, (unsigned long long)tot) : 0;
        }
        __syncthreads();
        if (stays) {
            u32 woff = 0;
            for (u32 w = 0; w < wave; ++w) woff += wcnt[w];
            const u64 at = bbase + woff + __popcll(mask & (lane ? (~0ull >> (64 - lane)) : 0ull));
            move_to[at] = (I)q; move_from[at] = (I)(base_pos + i);
        }
        __syncthreads();
    }
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// scratch that lives across the passes: re-allocated only when a pass needs more than any before it (the block sizes
// differ from pass to pass, so handing them back each time defeats the caching allocator and hipMalloc/hipFree of
// multi-GiB blocks cost 100s of ms under memory pressure)
int ensure(DevBuf& d, size_t bytes, hipStream_t stream) {
    if (d.bytes >= bytes && d.p) return KATOME_OK;
    return d.alloc(bytes + bytes / 8 + 256, stream);
}

int upload(DevBuf& d, const U32Buf& h, hipStream_t stream) {
    KCHECK(ensure(d, h.size() * 4 + 16, stream));
    if (!h.empty()) KCHECK_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, stream));
    return KATOME_OK;
}

}  // namespace

template <class I> __global__ void retain_jump_kernel(I* __restrict__ jump, u64 u, u64 M, u32* __restrict__ changed);

// remove_paths' edge removals (pruner.rs:199-217) for marked positions d_pos[u] (ascending) listed d_mult[] times each,
// of E edges: victims in removal order, the moves (to[i] <- from[i]) that fill the marked positions below the new count
template <class I>
static int replay_edges_t(const I* d_pos, const uint32_t* d_mult, uint64_t u, uint64_t E, ReplayScratch& sc, DevBuf& d_victims, DevBuf& to_e,
                          DevBuf& from_e, uint64_t* n_removed, uint64_t* n_left, uint64_t* n_moves, uint64_t* n_dups, hipStream_t stream) {
    constexpr size_t W = sizeof(I);
    *n_removed = 0; *n_left = E; *n_moves = 0; *n_dups = 0;
    if (u == 0) return KATOME_OK;
    DevBuf &agg = sc.agg, &carry = sc.carry, &size_before = sc.size_before, &jump = sc.jump, &totals = sc.totals;
    KCHECK(ensure(totals, 64, stream));
    KCHECK_HIP(hipMemsetAsync(totals.p, 0, 64, stream));
    const u64 nb = (u + (u64)BLOCK * REPLAY_ITEMS - 1) / ((u64)BLOCK * REPLAY_ITEMS);
    KCHECK(ensure(agg, nb * sizeof(Clamp) + 16, stream)); KCHECK(ensure(carry, nb * sizeof(Clamp) + 16, stream));
    KCHECK(ensure(size_before, u * W + 16, stream));
    hipLaunchKernelGGL((replay_scan_kernel<0, I>), dim3((unsigned)nb), dim3(BLOCK), 0, stream, d_pos, d_mult, u, E,
                       agg.as<Clamp>(), (const Clamp*)nullptr, (I*)nullptr, totals.as<u64>());
    hipLaunchKernelGGL(replay_spine_kernel, dim3(1), dim3(BLOCK), 0, stream, agg.as<Clamp>(), nb, carry.as<Clamp>());
    hipLaunchKernelGGL((replay_scan_kernel<1, I>), dim3((unsigned)nb), dim3(BLOCK), 0, stream, d_pos, d_mult, u, E,
                       (Clamp*)nullptr, carry.as<Clamp>(), size_before.as<I>(), totals.as<u64>());
    KCHECK_HIP(hipGetLastError());
    u64 left = 0;
    KCHECK_HIP(hipMemcpyAsync(&left, totals.as<u64>() + 3, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    const u64 E_new = left, m = E - E_new;
    KCHECK(ensure(jump, m * W + 16, stream));
    KCHECK(ensure(d_victims, m * W + 16, stream));
    KCHECK(ensure(to_e, u * W + 16, stream)); KCHECK(ensure(from_e, u * W + 16, stream));
    const u64 all = u;                                           // totals[4]: entries below the new count (default: all)
    KCHECK_HIP(hipMemcpyAsync(totals.as<u64>() + 4, &all, 8, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(iota_from_kernel<I>, dim3(grid_for(m, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, jump.as<I>(), m, (I)E_new);
    hipLaunchKernelGGL(replay_links_kernel<I>, dim3(grid_for(u, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, d_pos, d_mult,
                       size_before.as<I>(), u, E_new, jump.as<I>(), totals.as<u64>());
    for (int round = 0; round < 64; ++round) {
        KCHECK_HIP(hipMemsetAsync(totals.as<u64>() + 6, 0, 8, stream));
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(retain_jump_kernel<I>, dim3(grid_for(m, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, jump.as<I>(), m, E_new,
                               reinterpret_cast<u32*>(totals.as<u64>() + 6));
        u64 changed = 0;
        KCHECK_HIP(hipMemcpyAsync(&changed, totals.as<u64>() + 6, 8, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        if (!changed) break;
    }
    hipLaunchKernelGGL(replay_emit_kernel<I>, dim3(grid_for(u, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, d_pos, d_mult,
                       size_before.as<I>(), u, E, E_new, jump.as<I>(), d_victims.as<I>(), to_e.as<I>(), from_e.as<I>(),
                       totals.as<u64>());
    KCHECK_HIP(hipGetLastError());
    u64 h2[2] = {0, 0};
    KCHECK_HIP(hipMemcpyAsync(h2, totals.as<u64>() + 4, 16, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    *n_removed = m; *n_left = E_new; *n_moves = h2[0]; *n_dups = h2[1];
    return KATOME_OK;
}
int dev_replay_edges(const uint32_t* d_pos, const uint32_t* d_mult, uint64_t u, uint64_t E, ReplayScratch& sc, DevBuf& d_victims, DevBuf& to_e,
                     DevBuf& from_e, uint64_t* n_removed, uint64_t* n_left, uint64_t* n_moves, uint64_t* n_dups, hipStream_t stream) {
    return replay_edges_t<u32>(d_pos, d_mult, u, E, sc, d_victims, to_e, from_e, n_removed, n_left, n_moves, n_dups, stream);
}
// the same replay on 64-bit positions: a graph sharded over several GPUs has more than 2^32 edges (BASELINE config 5)
int dev_replay_edges64(const uint64_t* d_pos, const uint32_t* d_mult, uint64_t u, uint64_t E, ReplayScratch& sc, DevBuf& d_victims, DevBuf& to_e,
                       DevBuf& from_e, uint64_t* n_removed, uint64_t* n_left, uint64_t* n_moves, uint64_t* n_dups, hipStream_t stream) {
    return replay_edges_t<u64>(d_pos, d_mult, u, E, sc, d_victims, to_e, from_e, n_removed, n_left, n_moves, n_dups, stream);
}

// remove_single_node after every removed edge (pruner.rs:206-225) for die[2t], die[2t+1] (the endpoints edge removal t
// leaves without edges, REPLAY_NONE = stays), of N nodes: the moves to[i] <- from[i] of the tail nodes that stay.
// *fell_back = 1: a chain was too long for the device form; nothing was produced and the caller runs the host replay.
template <class I>
static int replay_nodes_t(const I* d_die, uint64_t m, uint64_t N, NodeReplayScratch& sc, DevBuf& to_n, DevBuf& from_n, uint64_t* n_moves,
                          uint64_t* n_left, int* fell_back, hipStream_t stream) {
    constexpr size_t W = sizeof(I);
    *n_moves = 0; *n_left = N; *fell_back = 0;
    if (m == 0) return KATOME_OK;
    const u64 nblocks = (m + (u64)BLOCK * MARK_ITEMS - 1) / ((u64)BLOCK * MARK_ITEMS);
    KCHECK(ensure(sc.counts, nblocks * 4 + 16, stream)); KCHECK(ensure(sc.offs, (nblocks + 1) * 8 + 16, stream));
    KCHECK(ensure(sc.flags, 32, stream));
    hipLaunchKernelGGL(die_count_kernel<I>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_die, m, sc.counts.as<u32>());
    KCHECK(dev_scan_counts(sc.counts.as<u32>(), nblocks, sc.offs.as<u64>(), stream));
    u64 M = 0;
    KCHECK_HIP(hipMemcpyAsync(&M, sc.offs.as<u64>() + nblocks, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    if (M == 0) return KATOME_OK;
    const u64 base_pos = N - M;
    KCHECK(ensure(sc.first, m * W + 16, stream)); KCHECK(ensure(sc.hole, M * W + 16, stream)); KCHECK(ensure(sc.dead, M + 16, stream));
    KCHECK(ensure(to_n, M * W + 16, stream)); KCHECK(ensure(from_n, M * W + 16, stream));
    KCHECK_HIP(hipMemsetAsync(sc.hole.p, 0xFF, M * W, stream));
    KCHECK_HIP(hipMemsetAsync(sc.dead.p, 0, M, stream));
    hipLaunchKernelGGL(die_first_kernel<I>, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_die, m, sc.offs.as<u64>(), base_pos, sc.first.as<I>(),
                       sc.dead.as<unsigned char>());
    u32 h_flags[2] = {1, 0};
    for (int round = 0; round < 256 && h_flags[0] && !h_flags[1]; ++round) {
        KCHECK_HIP(hipMemsetAsync(sc.flags.p, 0, 32, stream));
        hipLaunchKernelGGL(node_holes_kernel<I>, dim3(grid_for(m, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, d_die, sc.first.as<I>(), m, N, base_pos,
                           sc.hole.as<I>(), sc.flags.as<u32>());
        KCHECK_HIP(hipGetLastError());
        KCHECK_HIP(hipMemcpyAsync(h_flags, sc.flags.p, 8, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
    }
    if (h_flags[0] || h_flags[1]) { *fell_back = 1; return KATOME_OK; }
    KCHECK_HIP(hipMemsetAsync(sc.flags.p, 0, 32, stream));
    hipLaunchKernelGGL(node_moves_kernel<I>, dim3(grid_for(M, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, sc.dead.as<unsigned char>(), M, base_pos,
                       sc.hole.as<I>(), to_n.as<I>(), from_n.as<I>(), reinterpret_cast<u64*>(sc.flags.as<u32>() + 4), sc.flags.as<u32>());
    KCHECK_HIP(hipGetLastError());
    u32 h_end[6] = {0, 0, 0, 0, 0, 0};
    KCHECK_HIP(hipMemcpyAsync(h_end, sc.flags.p, 24, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    if (h_end[1]) { *fell_back = 1; return KATOME_OK; }
    *n_moves = (u64)h_end[4] | ((u64)h_end[5] << 32);
    *n_left = base_pos;
    return KATOME_OK;
}
int dev_replay_nodes(const uint32_t* d_die, uint64_t m, uint64_t N, NodeReplayScratch& sc, DevBuf& to_n, DevBuf& from_n, uint64_t* n_moves,
                     uint64_t* n_left, int* fell_back, hipStream_t stream) {
    return replay_nodes_t<u32>(d_die, m, N, sc, to_n, from_n, n_moves, n_left, fell_back, stream);
}
// (64-bit positions: the sharded graph's replay on rank 0, dist_prune.hip.  A pass whose node moves chain further than the device
// form follows -- or KATOME_PRUNE_HOST_NODES, as on one GPU -- is replayed by the sequential statement of prune_replay.h on the
// host instead: *fell_back = 1 then only reports that, the moves are produced either way)
int dev_replay_nodes64(const uint64_t* d_die, uint64_t m, uint64_t N, NodeReplayScratch& sc, DevBuf& to_n, DevBuf& from_n, uint64_t* n_moves,
                       uint64_t* n_left, int* fell_back, hipStream_t stream) {
    *fell_back = getenv("KATOME_PRUNE_HOST_NODES") != nullptr;
    if (!*fell_back) KCHECK(replay_nodes_t<u64>(d_die, m, N, sc, to_n, from_n, n_moves, n_left, fell_back, stream));
    if (!*fell_back) return KATOME_OK;
    PodBuf<uint64_t> h_die;
    h_die.need(2 * m + 1);
    if (!h_die.p) { set_error("out of host memory"); return KATOME_E_OOM; }
    if (m) KCHECK_HIP(hipMemcpyAsync(h_die.p, d_die, 2 * m * 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    NodeReplayT<uint64_t> nr;
    replay_nodes<uint64_t>(h_die.p, m, N, nr);
    *n_moves = nr.move_to.size(); *n_left = nr.n_new;
    KCHECK(to_n.alloc((*n_moves + 1) * 8)); KCHECK(from_n.alloc((*n_moves + 1) * 8));
    if (*n_moves) {
        KCHECK_HIP(hipMemcpyAsync(to_n.p, nr.move_to.p, *n_moves * 8, hipMemcpyHostToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(from_n.p, nr.move_from.p, *n_moves * 8, hipMemcpyHostToDevice, stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));           // (the host vectors go out of scope)
    return KATOME_OK;
}

int dev_remove_dead_paths(PruneGraph& g, uint32_t k, katome_prune_stats* st, hipStream_t stream) {
    katome_prune_stats local;
    memset(&local, 0, sizeof local);
    if (g.n_edges >= 0xFFFFFFFFull || g.n_nodes >= 0xFFFFFFFFull) {
        set_error("remove_dead_paths: more than 2^32 edges or nodes on one GPU");
        return KATOME_E_UNSUPPORTED;
    }
    const u32 nw = g.nw, two_k = 2 * k;
    u64 E = g.n_edges, N = g.n_nodes;
    u64* src = g.edge_src->as<u64>(); u64* dst = g.edge_dst->as<u64>();
    u32* weight = g.edge_weight->as<u32>(); u64* key = g.edge_key->as<u64>(); u64* node_key = g.node_key->as<u64>();
    DevBuf node_deg(stream), first_out(stream), mult(stream), totals(stream), touched(stream);
    DevBuf& orig = *g.edge_age;
    if (orig.bytes < (E + 1) * 4) {
        KCHECK(orig.alloc((E + 1) * 4, stream));
        KCHECK(dev_iota(orig.as<u32>(), E, stream));
    }
    KCHECK(node_deg.alloc((N + 1) * 8)); KCHECK(first_out.alloc((N + 1) * 8)); KCHECK(mult.alloc((E + 1) * 4));
    KCHECK(totals.alloc(64));
    PinnedU32 h_pos, h_mult, h_die;
    EdgeReplay er; NodeReplay nr;
    DevBuf counts(stream), offs(stream), d_pos(stream), d_mult(stream), d_victims(stream), last_touch(stream), d_die(stream);
    DevBuf to_e(stream), from_e(stream), to_n(stream), from_n(stream), tail_map(stream), inputs(stream);
    ReplayScratch replay_scratch(stream);
    NodeReplayScratch node_scratch(stream);
    const bool host_nodes = getenv("KATOME_PRUNE_HOST_NODES") != nullptr;
    const bool host_edges = getenv("KATOME_PRUNE_HOST_EDGES") != nullptr;      // the sequential replay of prune_replay.h (A/B checks)
    KCHECK(last_touch.alloc((N + 1) * 4));
    // adjacency slots (32 B per vertex): without them -- not enough memory, or KATOME_PRUNE_NO_SLOTS -- every pass streams all edges
    DevBuf out_slots(stream), in_slots(stream), redo_list(stream), stamp(stream);
    Slots slots{nullptr, nullptr, nw, k};
    if (!g.parallel_edges && !getenv("KATOME_PRUNE_NO_SLOTS") && out_slots.alloc((N + 1) * 16) == KATOME_OK && in_slots.alloc((N + 1) * 16) == KATOME_OK) {
        slots.out = out_slots.as<u32>(); slots.in = in_slots.as<u32>();
    } else {
        out_slots.release(); in_slots.release();
    }
    const bool trace = getenv("KATOME_TRACE_PRUNE") != nullptr;
    double lap_t = now_ms();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(stream);
        const double t = now_ms();
        fprintf(stderr, "[prune]   %-18s %9.2f ms\n", what, t - lap_t);
        lap_t = t;
    };
    // adjacency summary, once
    KCHECK_HIP(hipMemsetAsync(node_deg.p, 0, N * 8, stream));
    KCHECK_HIP(hipMemsetAsync(first_out.p, 0, N * 8, stream));
    KCHECK_HIP(hipMemsetAsync(mult.p, 0, E * 4, stream));
    KCHECK(touched.alloc((E >> MARK_GROUP_SHIFT) + 16));
    KCHECK_HIP(hipMemsetAsync(touched.p, 0, (E >> MARK_GROUP_SHIFT) + 16, stream));
    if (E) hipLaunchKernelGGL(degree_kernel, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, dst, orig.as<u32>(), E,
                              node_deg.as<u64>(), first_out.as<u64>());
    if (N) hipLaunchKernelGGL(successor_kernel, dim3(grid_for(N, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, N, first_out.as<u64>(), dst,
                              node_deg.as<u64>());
    KCHECK(stamp.alloc((slots.out ? N + 1 : 1) * 4));
    if (slots.out) KCHECK_HIP(hipMemsetAsync(stamp.p, 0, (N + 1) * 4, stream));
    if (slots.out && E) {
        KCHECK_HIP(hipMemsetAsync(out_slots.p, 0xFF, N * 16, stream));
        KCHECK_HIP(hipMemsetAsync(in_slots.p, 0xFF, N * 16, stream));
        hipLaunchKernelGGL(slots_init_kernel, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, dst, key, E, slots);
    }
    KCHECK_HIP(hipGetLastError());
    lap("adjacency");
    bool nodes_moved = false;                 // tail_map holds the moves of the pass before
    bool have_inputs = false;                 // `inputs` already holds this pass's Input vertices
    u64 n_inputs = 0;
    DevBuf inputs_next(stream), fresh_inputs(stream), changed(stream);
    const bool check_walks = getenv("KATOME_PRUNE_CHECK_WALKS") != nullptr;   // walk everything and verify that the skipped walks are not dead
    u32 walk_from = 0;                        // 0: walk every listed vertex; else only those stamped at or above this (affected_kernel)
    const u64 track_max = getenv("KATOME_PRUNE_WALK_ALL") ? 0 : (1ull << 20);   // passes that remove more edges than this re-walk everything
    while (E) {
        const double pass_t0 = now_ms();
        double pass_host = 0;
        // (1) walks
        KCHECK_HIP(hipMemsetAsync(totals.p, 0, 64, stream));
        if (have_inputs) {                                   // carried over from the pass before (inputs_update_kernel)
            KCHECK_HIP(hipMemcpyAsync(totals.as<u64>() + 2, &n_inputs, 8, hipMemcpyHostToDevice, stream));
        } else {
            KCHECK(ensure(inputs, N * 4 + 16, stream));
            hipLaunchKernelGGL(input_list_kernel, dim3(grid_for(N, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, N, node_deg.as<u64>(),
                               first_out.as<u64>(), dst, (nodes_moved && !slots.out) ? tail_map.as<u32>() : (const u32*)nullptr, inputs.as<u32>(), totals.as<u64>(), slots.out != nullptr);
        }
        hipLaunchKernelGGL(walk_kernel, dim3(256u * 16u), dim3(BLOCK), 0, stream, inputs.as<u32>(), two_k, first_out.as<u64>(), dst,
                           node_deg.as<u64>(), mult.as<u32>(), touched.as<unsigned char>(), stamp.as<u32>(), walk_from, check_walks, totals.as<u64>());
        KCHECK_HIP(hipGetLastError());
        u64 h_tot[5] = {0, 0, 0, 0, 0};
        KCHECK_HIP(hipMemcpyAsync(h_tot, totals.p, 40, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        lap("walks");
        if (h_tot[4]) { set_error("remove_dead_paths: %llu dead walks started at vertices the change tracking would have skipped (pass %llu)", (unsigned long long)h_tot[4], (unsigned long long)local.passes + 1); return KATOME_E_DEVICE; }
        local.passes += 1;
        local.walks += h_tot[3];
        if (h_tot[0] == 0) break;                         // to_remove.is_empty() (pruner.rs:76)
        local.dead_walks += h_tot[1];
        local.marked += h_tot[0];
        // (2) the marked indices, ascending (the reference sorts them descending; they are consumed from the top)
        const u64 nblocks = (E + (u64)BLOCK * MARK_ITEMS - 1) / ((u64)BLOCK * MARK_ITEMS);
        KCHECK(ensure(counts, nblocks * 4 + 16, stream)); KCHECK(ensure(offs, (nblocks + 1) * 8 + 16, stream));
        hipLaunchKernelGGL(mark_count_kernel, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, mult.as<u32>(), E, counts.as<u32>(),
                           (const unsigned char*)touched.as<unsigned char>());
        KCHECK(dev_scan_counts(counts.as<u32>(), nblocks, offs.as<u64>(), stream));
        u64 u = 0;
        KCHECK_HIP(hipMemcpyAsync(&u, offs.as<u64>() + nblocks, 8, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        KCHECK(ensure(d_pos, u * 4 + 16, stream)); KCHECK(ensure(d_mult, u * 4 + 16, stream));
        hipLaunchKernelGGL(mark_write_kernel, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, mult.as<u32>(), E, offs.as<u64>(),
                           d_pos.as<u32>(), d_mult.as<u32>(), (const unsigned char*)touched.as<unsigned char>());
        if (u) hipLaunchKernelGGL(mark_clear_kernel, dim3(grid_for(u, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, d_pos.as<u32>(), u, mult.as<u32>(),
                                  touched.as<unsigned char>());
        KCHECK_HIP(hipGetLastError());
        u64 m = 0, E_new = E, n_edge_moves = 0, dups = 0;
        double t_edges = 0;
        if (host_edges) {
            KCHECK(h_pos.need(u)); KCHECK(h_mult.need(u));
            KCHECK_HIP(hipMemcpyAsync(h_pos.p, d_pos.p, u * 4, hipMemcpyDeviceToHost, stream));
            KCHECK_HIP(hipMemcpyAsync(h_mult.p, d_mult.p, u * 4, hipMemcpyDeviceToHost, stream));
            KCHECK_HIP(hipStreamSynchronize(stream));
            lap("marks to host");
            // (3) replay of remove_edge
            const double t0 = now_ms();
            replay_edges(h_pos.p, h_mult.p, u, E, h_tot[0], er);
            t_edges = now_ms() - t0;
            local.host_ms += t_edges; pass_host += t_edges;
            m = er.victims.size(); E_new = er.n_new; dups = er.from_duplicates; n_edge_moves = er.move_to.size();
            KCHECK(upload(d_victims, er.victims, stream));
            KCHECK(upload(to_e, er.move_to, stream)); KCHECK(upload(from_e, er.move_from, stream));
        } else {
            // (3) the same replay as a scan + pointer jumping, without leaving the device
            KCHECK(dev_replay_edges(d_pos.as<u32>(), d_mult.as<u32>(), u, E, replay_scratch, d_victims, to_e, from_e, &m, &E_new, &n_edge_moves, &dups, stream));
        }
        local.removed_edges += m;
        local.removed_by_duplicates += dups;
        lap("replay edges");
        // (4) the nodes each removal isolates
        KCHECK(ensure(d_die, 2 * m * 4 + 16, stream));
        KCHECK_HIP(hipMemsetAsync(last_touch.p, 0, N * 4, stream));
        if (slots.out) {
            KCHECK(ensure(redo_list, m * 4 + 16, stream)); KCHECK(ensure(fresh_inputs, 2 * m * 4 + 16, stream));
            KCHECK_HIP(hipMemsetAsync(totals.as<u64>() + 4, 0, 24, stream));       // [4] changed vertices, [5] lost their first out-edge, [6] new Inputs
        }
        const bool track = slots.out && m <= track_max;
        if (track) KCHECK(ensure(changed, 2 * m * 4 + 16, stream));
        hipLaunchKernelGGL(death_count_kernel, dim3(grid_for(m, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, d_victims.as<u32>(), m, src, dst,
                           node_deg.as<u64>(), first_out.as<u64>(), last_touch.as<u32>(), slots, key, redo_list.as<u32>(), totals.as<u64>() + 5);
        hipLaunchKernelGGL(death_emit_kernel, dim3(grid_for(m, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, d_victims.as<u32>(), m, src, dst,
                           node_deg.as<u64>(), last_touch.as<u32>(), d_die.as<u32>(), slots.out ? fresh_inputs.as<u32>() : (u32*)nullptr,
                           totals.as<u64>() + 6, track ? changed.as<u32>() : (u32*)nullptr, totals.as<u64>() + 4);
        KCHECK_HIP(hipGetLastError());
        // (5) replay of remove_node: on the device, or (chains too long for that form, or asked for) on the host
        u64 n_node_moves = 0, N_new = N;
        double t_nodes = 0;
        int on_host = host_nodes ? 1 : 0;
        if (!on_host) KCHECK(dev_replay_nodes(d_die.as<u32>(), m, N, node_scratch, to_n, from_n, &n_node_moves, &N_new, &on_host, stream));
        if (on_host) {
            KCHECK(h_die.need(2 * m));
            KCHECK_HIP(hipMemcpyAsync(h_die.p, d_die.p, 2 * m * 4, hipMemcpyDeviceToHost, stream));
            KCHECK_HIP(hipStreamSynchronize(stream));
            lap("deaths to host");
            const double t0 = now_ms();
            replay_nodes(h_die.p, m, N, nr);
            t_nodes = now_ms() - t0;
            local.host_ms += t_nodes; pass_host += t_nodes;
            n_node_moves = nr.move_to.size(); N_new = nr.n_new;
            KCHECK(upload(to_n, nr.move_to, stream)); KCHECK(upload(from_n, nr.move_from, stream));
        }
        local.removed_nodes += N - N_new;
        lap("replay nodes");
        // (6) apply the moves, re-label the endpoints of the surviving edges
        {
            const u64 ne = n_edge_moves;
            if (ne) hipLaunchKernelGGL(move_edges_kernel, dim3(grid_for(ne, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, to_e.as<u32>(),
                                       from_e.as<u32>(), ne, nw, src, dst, weight, orig.as<u32>(), key, first_out.as<u64>(), slots);
            E = E_new;
            const u64 nn = n_node_moves;
            nodes_moved = nn != 0;
            KCHECK(ensure(tail_map, (N - N_new + 1) * 4, stream));
            if (slots.out) {
                // what changed, and nothing else: new first out-edges (edge positions are final, node ids still the old ones),
                // then the vertices move, then the edges in a moved vertex's slots learn its new id
                u64 h_cnt[3] = {0, 0, 0};
                KCHECK_HIP(hipMemcpyAsync(h_cnt, totals.as<u64>() + 4, 24, hipMemcpyDeviceToHost, stream));
                KCHECK_HIP(hipStreamSynchronize(stream));
                const u64 n_changed = h_cnt[0], n_redo = h_cnt[1], n_fresh = h_cnt[2];
                if (n_redo) hipLaunchKernelGGL(redo_slots_kernel, dim3(grid_for(n_redo, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, redo_list.as<u32>(),
                                               n_redo, slots, orig.as<u32>(), dst, node_deg.as<u64>(), first_out.as<u64>());
                if (nn) hipLaunchKernelGGL(move_nodes_kernel, dim3(grid_for(nn, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, to_n.as<u32>(),
                                           from_n.as<u32>(), nn, nw, N_new, node_key, node_deg.as<u64>(), first_out.as<u64>(), tail_map.as<u32>(), slots);
                if (nn) hipLaunchKernelGGL(remap_slots_kernel, dim3(grid_for(nn, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, to_n.as<u32>(), nn, N_new,
                                           tail_map.as<u32>(), slots, src, dst, first_out.as<u64>(), node_deg.as<u64>());
                // next pass's Input vertices from this pass's, without a look at every vertex
                const u64 n_old = h_tot[2], cand = n_old + n_fresh;
                KCHECK(ensure(inputs_next, cand * 4 + 16, stream));
                KCHECK_HIP(hipMemsetAsync(totals.as<u64>() + 7, 0, 8, stream));
                if (cand) hipLaunchKernelGGL(inputs_update_kernel, dim3(grid_for(cand, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, inputs.as<u32>(), n_old,
                                             fresh_inputs.as<u32>(), n_fresh, last_touch.as<u32>(), N_new, tail_map.as<u32>(), node_deg.as<u64>(),
                                             inputs_next.as<u32>(), totals.as<u64>() + 7, stamp.as<u32>(), (u32)(local.passes + 1) * STAMP_STEP + (STAMP_STEP - 1));
                // which of them the next pass has to walk: everything, or what lies within reach of a change
                walk_from = 0;
                if (track) {
                    if (n_changed) hipLaunchKernelGGL(affected_kernel, dim3(grid_for(n_changed, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, changed.as<u32>(),
                                                      n_changed, N_new, tail_map.as<u32>(), slots, src, node_deg.as<u64>(), stamp.as<u32>(),
                                                      (u32)(local.passes + 1) * STAMP_STEP + (STAMP_STEP - 1), two_k);
                    walk_from = (u32)(local.passes + 1) * STAMP_STEP;
                }
                KCHECK_HIP(hipGetLastError());
                KCHECK_HIP(hipMemcpyAsync(&n_inputs, totals.as<u64>() + 7, 8, hipMemcpyDeviceToHost, stream));
                KCHECK_HIP(hipStreamSynchronize(stream));
                std::swap(inputs.p, inputs_next.p); std::swap(inputs.bytes, inputs_next.bytes);
                have_inputs = true;
            } else {
                if (nn) hipLaunchKernelGGL(move_nodes_kernel, dim3(grid_for(nn, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, to_n.as<u32>(),
                                           from_n.as<u32>(), nn, nw, N_new, node_key, node_deg.as<u64>(), first_out.as<u64>(), tail_map.as<u32>(),
                                           Slots{nullptr, nullptr, 0, 0});
                if (E && nn) hipLaunchKernelGGL(first_out_redo_kernel<true>, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, dst,
                                                orig.as<u32>(), E, N_new, tail_map.as<u32>(), node_deg.as<u64>(), first_out.as<u64>());
                else if (E) hipLaunchKernelGGL(first_out_redo_kernel<false>, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, dst,
                                               orig.as<u32>(), E, N_new, (const u32*)nullptr, node_deg.as<u64>(), first_out.as<u64>());
            }
            KCHECK_HIP(hipGetLastError());
            N = N_new;
            KCHECK_HIP(hipStreamSynchronize(stream));      // the host vectors are reused by the next pass
        }
        lap("apply");
        if (trace)
            fprintf(stderr, "[prune] pass %llu: E %llu N %llu walks %llu dead %llu marked %llu removed %llu (dup %llu) nodes %llu | %.2f ms, host %.2f (edges %.2f nodes %.2f)\n",
                    (unsigned long long)local.passes, (unsigned long long)E, (unsigned long long)N, (unsigned long long)h_tot[2],
                    (unsigned long long)h_tot[1], (unsigned long long)h_tot[0], (unsigned long long)m,
                    (unsigned long long)dups, (unsigned long long)n_node_moves, now_ms() - pass_t0, pass_host, t_edges, t_nodes);
    }
    g.n_edges = E; g.n_nodes = N;
    if (st) *st = local;
    return KATOME_OK;
}

// ---- retain_edges / retain_nodes on the device ---------------------------------------------------------------------
// petgraph's retain_* visit the indices in descending order and swap_remove the rejected ones; every index is listed
// once, and then the replay has a parallel form.  With u of n indices flagged, the j-th removal (descending) happens
// when position n - j is the last one, and what sits there moves into the removed index.  So for a flagged index p with
// c flagged indices below it (j = u - c): next(p) = n - (u - c) is where its new occupant comes from, and that
// occupant is whatever ended up at next(p) before -- itself if next(p) is not flagged, else recursively the occupant of
// next(next(p)) (always a larger position).  The flagged indices under the new count M = n - u are the holes that get
// filled: hole p <- f(next(p)), f followed to an unflagged position by pointer jumping over the tail [M, n).
__global__ __launch_bounds__(BLOCK) void retain_links_kernel(const u32* __restrict__ flag, u64 n, u64 u, const u64* __restrict__ block_offs,
                                                             u32* __restrict__ jump /* [u]: tail position q -> jump[q - M] */,
                                                             u32* __restrict__ hole_to, u32* __restrict__ hole_from) {
    __shared__ u32 wsum[BLOCK / 64];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 base = ((u64)blockIdx.x * BLOCK + tid) * MARK_ITEMS, M = n - u;
    bool f[MARK_ITEMS]; u32 c = 0;
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j) { f[j] = base + j < n && flag[base + j] != 0; c += f[j]; }
    u32 incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 woff = 0;
    for (u32 w = 0; w < wave; ++w) woff += wsum[w];
    u64 before = block_offs[blockIdx.x] + woff + incl - c;          // flagged indices below base
#pragma unroll
    for (int j = 0; j < MARK_ITEMS; ++j) {
        const u64 p = base + j;
        if (p >= n) break;
        if (f[j]) {
            const u64 next = n - (u - before);
            if (p >= M) jump[p - M] = (u32)next;
            else { hole_to[before] = (u32)p; hole_from[before] = (u32)next; }   // the holes are the first flagged indices
            ++before;
        } else if (p >= M) {
            jump[p - M] = (u32)p;
        }
    }
}
template <class I>
__global__ __launch_bounds__(BLOCK) void retain_jump_kernel(I* __restrict__ jump, u64 u, u64 M, u32* __restrict__ changed) {
    bool any = false;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < u; i += (u64)gridDim.x * BLOCK) {
        const I v = jump[i];
        if (v == (I)(M + i)) continue;                             // not flagged, or removed while last: a fixed point
        const I w = jump[v - M];
        if (w != v) { jump[i] = w; any = true; }
    }
    if (any) *changed = 1;
}
__global__ __launch_bounds__(BLOCK) void retain_resolve_kernel(const u32* __restrict__ jump, u64 M, u64 holes, u32* __restrict__ hole_from) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < holes; i += (u64)gridDim.x * BLOCK) hole_from[i] = jump[hole_from[i] - M];
}

// flag[i] != 0: index i is rejected.  -> moves (to[k] <- from[k], k < *n_moves) on the device and the new count
static int retain_on_device(const u32* d_flag, u64 n, DevBuf& to, DevBuf& from, u64* n_moves, u64* n_new, hipStream_t stream) {
    const u64 nblocks = (n + (u64)BLOCK * MARK_ITEMS - 1) / ((u64)BLOCK * MARK_ITEMS);
    DevBuf counts(stream), offs(stream), jump(stream), changed(stream);
    KCHECK(counts.alloc(nblocks * 4 + 16)); KCHECK(offs.alloc((nblocks + 1) * 8 + 16)); KCHECK(changed.alloc(16));
    hipLaunchKernelGGL(mark_count_kernel, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_flag, n, counts.as<u32>(), (const unsigned char*)nullptr);
    KCHECK(dev_scan_counts(counts.as<u32>(), nblocks, offs.as<u64>(), stream));
    u64 u = 0;
    KCHECK_HIP(hipMemcpyAsync(&u, offs.as<u64>() + nblocks, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    const u64 M = n - u;
    *n_new = M; *n_moves = 0;
    if (u == 0 || M == 0) return KATOME_OK;
    // how many holes: flagged indices below M.  Everything flagged is either a hole or in the tail; the tail holds u
    // positions of which (u - holes) are flagged, so holes = unflagged tail positions = survivors that move
    KCHECK(jump.alloc(u * 4 + 16));
    KCHECK(ensure(to, std::min(u, M) * 4 + 16, stream)); KCHECK(ensure(from, std::min(u, M) * 4 + 16, stream));
    hipLaunchKernelGGL(retain_links_kernel, dim3((unsigned)nblocks), dim3(BLOCK), 0, stream, d_flag, n, u, offs.as<u64>(), jump.as<u32>(),
                       to.as<u32>(), from.as<u32>());
    // holes = flagged below M = prefix count at M: read it off the block offsets is not exact, so count it from the scan
    // of the block that holds M: simpler, ask the device for the number of flagged indices below M
    u64 holes = 0;
    {
        // flagged below M = u - flagged in [M, n); the tail is u positions long: count its flagged ones with the same kernels
        DevBuf tcounts(stream), toffs(stream);
        const u64 tblocks = (u + (u64)BLOCK * MARK_ITEMS - 1) / ((u64)BLOCK * MARK_ITEMS);
        KCHECK(tcounts.alloc(tblocks * 4 + 16)); KCHECK(toffs.alloc((tblocks + 1) * 8 + 16));
        hipLaunchKernelGGL(mark_count_kernel, dim3((unsigned)tblocks), dim3(BLOCK), 0, stream, d_flag + M, u, tcounts.as<u32>(), (const unsigned char*)nullptr);
        KCHECK(dev_scan_counts(tcounts.as<u32>(), tblocks, toffs.as<u64>(), stream));
        u64 in_tail = 0;
        KCHECK_HIP(hipMemcpyAsync(&in_tail, toffs.as<u64>() + tblocks, 8, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        holes = u - in_tail;
    }
    // pointer jumping: chains only run upwards and end at an unflagged position
    for (int round = 0; round < 64; ++round) {
        KCHECK_HIP(hipMemsetAsync(changed.p, 0, 4, stream));
        for (int rep = 0; rep < 2; ++rep)
            hipLaunchKernelGGL(retain_jump_kernel, dim3(grid_for(u, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, jump.as<u32>(), u, M, changed.as<u32>());
        u32 h = 0;
        KCHECK_HIP(hipMemcpyAsync(&h, changed.p, 4, hipMemcpyDeviceToHost, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        if (!h) break;
    }
    if (holes) hipLaunchKernelGGL(retain_resolve_kernel, dim3(grid_for(holes, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, jump.as<u32>(), M, holes, from.as<u32>());
    KCHECK_HIP(hipGetLastError());
    *n_moves = holes;
    return KATOME_OK;
}

// Clean::remove_weak_edges for PtGraph (pruner.rs:84-93) with petgraph's numbering: retain_edges visits the edge
// indices in descending order and swap_removes those below the threshold, then retain_nodes does the same with the
// nodes left without neighbours -- the replay of prune_replay.h with every index listed once, for edges and for nodes.
int dev_remove_weak_edges_ordered(PruneGraph& g, uint32_t threshold, hipStream_t stream) {
    if (g.n_edges >= 0xFFFFFFFFull || g.n_nodes >= 0xFFFFFFFFull) {
        set_error("remove_weak_edges: more than 2^32 edges or nodes on one GPU");
        return KATOME_E_UNSUPPORTED;
    }
    const u32 nw = g.nw;
    u64 E = g.n_edges, N = g.n_nodes;
    if (E == 0) return KATOME_OK;
    u64* src = g.edge_src->as<u64>(); u64* dst = g.edge_dst->as<u64>();
    u32* weight = g.edge_weight->as<u32>(); u64* key = g.edge_key->as<u64>(); u64* node_key = g.node_key->as<u64>();
    DevBuf flag(stream), to(stream), from(stream);
    KCHECK(flag.alloc((std::max(E, N) + 1) * 4));
    DevBuf& orig = *g.edge_age;                            // the edges' ages move with them: remove_dead_paths may follow
    if (orig.bytes < (E + 1) * 4) {
        KCHECK(orig.alloc((E + 1) * 4, stream));
        KCHECK(dev_iota(orig.as<u32>(), E, stream));
    }
    // retain_edges(|e| weight >= threshold)
    hipLaunchKernelGGL(weak_flag_kernel, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, weight, E, threshold, flag.as<u32>());
    u64 n_moves = 0, E_new = E;
    KCHECK(retain_on_device(flag.as<u32>(), E, to, from, &n_moves, &E_new, stream));
    if (n_moves)
        hipLaunchKernelGGL(move_edges_kernel, dim3(grid_for(n_moves, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, to.as<u32>(),
                           from.as<u32>(), n_moves, nw, src, dst, weight, orig.as<u32>(), key, (u64*)nullptr, Slots{nullptr, nullptr, 0, 0});
    E = E_new;
    // retain_nodes(|n| has a neighbour)
    hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(N, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, flag.as<u32>(), N, 1u);
    if (E) hipLaunchKernelGGL(touch_nodes_kernel, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, dst, E, flag.as<u32>());
    KCHECK_HIP(hipGetLastError());
    u64 nn = 0, N_new = N;
    KCHECK(retain_on_device(flag.as<u32>(), N, to, from, &nn, &N_new, stream));
    DevBuf tail_map(stream);
    KCHECK(tail_map.alloc((N - N_new + 1) * 4));
    if (nn) hipLaunchKernelGGL(move_nodes_kernel, dim3(grid_for(nn, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, to.as<u32>(), from.as<u32>(),
                               nn, nw, N_new, node_key, (u64*)nullptr, (u64*)nullptr, tail_map.as<u32>(), Slots{nullptr, nullptr, 0, 0});
    if (E && nn) hipLaunchKernelGGL(remap_kernel, dim3(grid_for(E, BLOCK, 256u * 32u)), dim3(BLOCK), 0, stream, src, dst, E, N_new, tail_map.as<u32>());
    KCHECK_HIP(hipGetLastError());
    KCHECK_HIP(hipStreamSynchronize(stream));
    g.n_edges = E; g.n_nodes = N_new;
    return KATOME_OK;
}

}  // namespace katome
