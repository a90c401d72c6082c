// prune_replay.h -- the two sequential replays of petgraph 0.4.13's swap_remove bookkeeping that
// Prunable::remove_dead_paths implies (reference pruner.rs:199-225 over Graph::remove_edge / remove_node).
// Host-only, no HIP: included by prune.hip and by tests/hostshim.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace katome {

constexpr uint32_t REPLAY_NONE = 0xFFFFFFFFu;

// grow-only array of u32 without value-initialisation: the first pass of a big graph fills hundreds of MiB, and touching
// that memory twice (zero-fill, then the real values) or copying it while a vector doubles costs as much as the replay
template <typename T>
struct PodBuf {
    T* p = nullptr;
    size_t cap = 0, n = 0;
    PodBuf() {}
    PodBuf(const PodBuf&) = delete;
    PodBuf& operator=(const PodBuf&) = delete;
    ~PodBuf() { free(p); }
    void need(size_t c) {                  // contents are NOT kept
        if (c > cap) { free(p); p = (T*)malloc((c + c / 8 + 64) * sizeof(T)); cap = p ? c + c / 8 + 64 : 0; }
        n = 0;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const T* data() const { return p; }
    T operator[](size_t i) const { return p[i]; }
};
typedef PodBuf<uint32_t> U32Buf;

// Edges (remove_paths, pruner.rs:199-217): the collected indices arrive ascending with multiplicities and are
// consumed from the top, as the reference's descending sort does.  remove_edge(d) moves the edge at the last
// position into d, so an index listed twice removes whatever was moved in.  The entry being removed and the last
// position both only move down, so the occupants that differ from the identity live in an array aligned with the
// entries and everything streams.
struct EdgeReplay {
    U32Buf victims;                    // identity (position at the start of the pass) of each removed edge, in order
    U32Buf move_to, move_from;
    U32Buf occ;
    uint64_t n_new = 0, from_duplicates = 0;
};
inline void replay_edges(const uint32_t* pos, const uint32_t* mult, uint64_t u, uint64_t n_edges, uint64_t marks, EdgeReplay& out) {
    out.occ.need(u);                                   // occupant of position pos[j]
    if (u) memcpy(out.occ.p, pos, u * 4);
    out.victims.need(marks);
    uint32_t* occ = out.occ.p;
    uint32_t* victims = out.victims.p;
    uint64_t size = n_edges, nv = 0, dups = 0;
    long long q = (long long)u - 1;
    for (long long j = (long long)u - 1; j >= 0; --j) {
        const uint32_t d = pos[j], c = mult[j];
        for (uint32_t r = 0; r < c; ++r) {
            if (d >= size) break;                      // edge_endpoints(e) == None and remove_edge(e) == None
            const uint32_t last = (uint32_t)(size - 1);
            while (q >= 0 && pos[q] > last) --q;
            const uint32_t mover = (q >= 0 && pos[q] == last) ? occ[q] : last;
            victims[nv++] = occ[j];
            dups += r != 0;
            if (d != last) occ[j] = mover;
            --size;
        }
    }
    out.victims.n = nv;
    out.n_new = size;
    out.from_duplicates = dups;
    uint64_t below = 0;                                // entries under the new count: at most that many moves
    while (below < u && pos[below] < size) ++below;
    out.move_to.need(below); out.move_from.need(below);
    uint64_t nm = 0;
    for (uint64_t j = 0; j < below; ++j)
        if (occ[j] != pos[j]) { out.move_to.p[nm] = pos[j]; out.move_from.p[nm] = occ[j]; ++nm; }
    out.move_to.n = out.move_from.n = nm;
}

// Nodes (remove_single_node after every removed edge, pruner.rs:206-225): die[2t], die[2t+1] name the endpoints
// (source, target) that removal t leaves without edges (REPLAY_NONE = stays); when both go, the one with the larger
// CURRENT index goes first.  remove_node moves the last node into the freed index, so only nodes of the tail that
// disappears are ever re-labelled: two arrays over that tail hold the whole state.
// (I = uint32_t on one GPU; uint64_t for the sharded graph, whose indices pass 2^32 -- dist_prune.hip's fall-back)
template <typename I>
struct NodeReplayT {
    PodBuf<I> move_to, move_from;
    PodBuf<I> tail_pos, tail_occ;
    uint64_t n_new = 0;
};
typedef NodeReplayT<uint32_t> NodeReplay;
template <typename I>
inline void replay_nodes(const I* die, uint64_t m, uint64_t n_nodes, NodeReplayT<I>& out) {
    const I REPLAY_NONE = (I)~(I)0;
    uint64_t n_die = 0;
    for (uint64_t i = 0; i < 2 * m; ++i) n_die += die[i] != REPLAY_NONE;
    const uint64_t base = n_nodes - n_die;
    out.tail_pos.need(n_die); out.tail_occ.need(n_die);
    I* tail_pos = out.tail_pos.p;               // current index of tail node base+i (REPLAY_NONE once removed)
    I* tail_occ = out.tail_occ.p;               // node at tail index base+i
    for (uint64_t i = 0; i < n_die; ++i) tail_pos[i] = tail_occ[i] = (I)(base + i);
    uint64_t size = n_nodes;
    auto pos_of = [&](I v) -> I { return v < base ? v : tail_pos[v - base]; };
    auto remove = [&](I v, I p) {
        const I top = (I)(size - 1), y = tail_occ[top - base];
        if (p != top) {
            if (p >= base) tail_occ[p - base] = y;
            tail_pos[y - base] = p;
        }
        if (v >= base) tail_pos[v - base] = REPLAY_NONE;
        --size;
    };
    for (uint64_t t = 0; t < m; ++t) {
        const I a = die[2 * t], b = die[2 * t + 1];
        if (a != REPLAY_NONE && b != REPLAY_NONE) {
            const I pa = pos_of(a), pb = pos_of(b);
            if (pa < pb) { remove(b, pb); remove(a, pos_of(a)); } else { remove(a, pa); remove(b, pos_of(b)); }
        } else if (a != REPLAY_NONE) {
            remove(a, pos_of(a));
        } else if (b != REPLAY_NONE) {
            remove(b, pos_of(b));
        }
    }
    out.n_new = size;
    out.move_to.need(n_die); out.move_from.need(n_die);
    uint64_t nm = 0;
    for (uint64_t i = 0; i < n_die; ++i)
        if (tail_pos[i] != REPLAY_NONE) { out.move_to.p[nm] = tail_pos[i]; out.move_from.p[nm] = (I)(base + i); ++nm; }
    out.move_to.n = out.move_from.n = nm;
}

}  // namespace katome
