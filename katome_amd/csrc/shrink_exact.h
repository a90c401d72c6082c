// shrink_exact.h -- Shrinkable::shrink for PtGraph exactly as the reference computes it (src/katome/algorithms/shrinker.rs:38-209):
// which edges end up merged (the cuts its traversal makes on tangled graphs and on cycles) AND the edge / node indices petgraph
// 0.4.13 leaves behind.  Both are functions of an inherently sequential process -- a depth-first traversal (ShrinkTraverse,
// shrinker.rs:62-135) interleaved with Graph::remove_edge (= Vec::swap_remove plus re-linking) and Graph::add_edge calls that change
// the adjacency it walks -- so this is a sequential statement over petgraph's own data layout: per node the heads of its outgoing /
// incoming edge lists, per edge its end points and the next edge of each list (newest edge first).  What is data-parallel stays on the
// device (shrink.hip): the adjacency order (edge ages), and the bytes -- every merged edge's label (EdgeSlice::merge, slices.rs:23-34)
// is written by following the chain of original edges this file hands back.
// Host-only, no HIP: included by shrink.hip and by tests/hostshim (checked there against the oracle on hand-made graphs).
#pragma once
#include <stdint.h>

#include <vector>

namespace katome {

struct ShrinkExact {
    static constexpr uint32_t END = 0xFFFFFFFFu;
    // ---- petgraph's layout (graph_impl: Node { next: [EdgeIndex; 2] }, Edge { next: [EdgeIndex; 2], node: [NodeIndex; 2] }) ----
    std::vector<uint32_t> node_next[2];           // [0] first outgoing edge, [1] first incoming edge
    std::vector<uint32_t> edge_next[2], edge_node[2];
    std::vector<uint32_t> edge_slot;              // the edge's weight (EdgeSlice, EdgeWeight) named by the ORIGINAL edge whose slot it holds
    uint32_t n_edges = 0, n_nodes = 0;
    // every slot's chain of original edges: EdgeSlice::merge appends the other edge's remainder, i.e. its chain
    std::vector<uint32_t> chain_next, chain_last;
    // ---- ShrinkTraverse (shrinker.rs:38-46) ----
    std::vector<uint64_t> fb;                      // visited bits
    std::vector<uint32_t> stack;
    uint64_t node_offset = 0;
    // ---- statistics ----
    uint64_t calls = 0, merges = 0, restarts = 0;

    bool visited(uint32_t n) const { return (fb[n >> 6] >> (n & 63)) & 1; }
    void mark(uint32_t n) { fb[n >> 6] |= 1ull << (n & 63); }
    bool degree_is_one(uint32_t n, int d) const { const uint32_t h = node_next[d][n]; return h != END && edge_next[d][h] == END; }
    bool isolated(uint32_t n) const { return node_next[0][n] == END && node_next[1][n] == END; }

    // the graph as it stands when shrink is called: edge e = (src[e], dst[e]); `by_age` lists the edges oldest first -- add_edge puts
    // an edge at the head of both lists (graph_impl add_edge), so the lists come out newest first, whatever swap_removes have done to
    // the indices since
    void init(const uint32_t* src, const uint32_t* dst, const uint32_t* by_age, uint32_t E, uint32_t N) {
        n_edges = E; n_nodes = N;
        for (int d = 0; d < 2; ++d) { node_next[d].assign(N, END); edge_next[d].resize(E); edge_node[d].resize(E); }
        edge_slot.resize(E); chain_next.assign(E, END); chain_last.resize(E);
        for (uint32_t e = 0; e < E; ++e) { edge_node[0][e] = src[e]; edge_node[1][e] = dst[e]; edge_slot[e] = e; chain_last[e] = e; }
        for (uint32_t i = 0; i < E; ++i) {
            const uint32_t e = by_age ? by_age[i] : i;
            for (int d = 0; d < 2; ++d) { const uint32_t n = edge_node[d][e]; edge_next[d][e] = node_next[d][n]; node_next[d][n] = e; }
        }
        // ShrinkTraverse::new (shrinker.rs:48-60): graph.externals(Incoming), ascending; the bit set has node_count() bits
        fb.assign(((uint64_t)N + 63) / 64, 0);
        stack.clear();
        for (uint32_t n = 0; n < N; ++n) if (node_next[1][n] == END) stack.push_back(n);
        node_offset = 0;
    }

    // Graph::change_edge_links: wherever `e` is linked in the two lists of its end points, link `to[k]` instead
    void change_edge_links(const uint32_t node[2], uint32_t e, const uint32_t to[2]) {
        for (int k = 0; k < 2; ++k) {
            const uint32_t fst = node_next[k][node[k]];
            if (fst == e) { node_next[k][node[k]] = to[k]; continue; }
            for (uint32_t cur = fst; cur != END; cur = edge_next[k][cur])
                if (edge_next[k][cur] == e) { edge_next[k][cur] = to[k]; break; }
        }
    }
    // Graph::remove_edge (+ remove_edge_adjust_indices): unlink, swap_remove, re-link the edge that moved in
    void remove_edge(uint32_t e) {
        const uint32_t node[2] = {edge_node[0][e], edge_node[1][e]}, next[2] = {edge_next[0][e], edge_next[1][e]};
        change_edge_links(node, e, next);
        const uint32_t last = --n_edges;
        if (e == last) return;
        for (int d = 0; d < 2; ++d) { edge_node[d][e] = edge_node[d][last]; edge_next[d][e] = edge_next[d][last]; }
        edge_slot[e] = edge_slot[last];
        const uint32_t swap[2] = {edge_node[0][e], edge_node[1][e]}, to[2] = {e, e};
        change_edge_links(swap, last, to);
    }
    uint32_t add_edge(uint32_t a, uint32_t b, uint32_t slot) {
        const uint32_t e = n_edges++;
        edge_node[0][e] = a; edge_node[1][e] = b; edge_slot[e] = slot;
        edge_next[0][e] = node_next[0][a]; edge_next[1][e] = node_next[1][b];
        node_next[0][a] = e; node_next[1][b] = e;
        return e;
    }

    // ShrinkTraverse::next (shrinker.rs:62-135); END = None
    uint32_t next() {
        for (;;) {
            while (!stack.empty()) {
                uint32_t current = stack.back();
                for (;;) {
                    bool new_ancestor = false;
                    for (uint32_t e = node_next[0][current]; e != END; e = edge_next[0][e]) {
                        const uint32_t n = edge_node[1][e];
                        if (visited(n)) continue;
                        mark(n);
                        if (current == n) continue;
                        if (degree_is_one(n, 0) && degree_is_one(n, 1)) return e;         // current -> n -> x: start shrinking here
                        stack.push_back(n); current = n; new_ancestor = true;
                        break;
                    }
                    if (!new_ancestor) { stack.pop_back(); mark(current); break; }
                }
            }
            // A component with a cycle at its root: "the next unvisited node", found as self.fb.zeros().skip(self.node_offset) -- the
            // offset counts the isolated nodes earlier scans passed over (and marked), so it also skips that many nodes that are
            // still unvisited: reproduced as written.  The isolated nodes a scan passes are marked once it has found a node to start
            // from; they are exactly the zeros between the skipped ones and that node, so no list of them is kept.
            uint64_t skip = node_offset, passed = 0;
            uint32_t start = END;
            uint64_t w0 = 0, first_bit_mask = 0;         // where the marking begins: word w0, only the bits in first_bit_mask
            bool began = false;
            const uint64_t n_words = fb.size();
            for (uint64_t w = 0; w < n_words && start == END; ++w) {
                uint64_t z = ~fb[w];
                if (w == n_words - 1 && (n_nodes & 63)) z &= (1ull << (n_nodes & 63)) - 1;
                if (!z) continue;
                const uint64_t c = (uint64_t)__builtin_popcountll(z);
                if (skip >= c) { skip -= c; continue; }
                while (skip) { z &= z - 1; --skip; }
                if (!began) { began = true; w0 = w; first_bit_mask = z; }
                for (uint64_t zz = z; zz; zz &= zz - 1) {
                    const uint32_t n = (uint32_t)(w * 64 + (uint64_t)__builtin_ctzll(zz));
                    if (isolated(n)) { ++passed; continue; }
                    start = n;
                    break;
                }
            }
            if (start == END) return END;
            // mark the isolated nodes passed: every zero from the first unskipped one up to (not including) `start`
            {
                const uint64_t ws = start >> 6;
                const uint64_t below = (1ull << (start & 63)) - 1;            // bits under `start` in its word
                if (w0 == ws) fb[ws] |= first_bit_mask & below;
                else {
                    fb[w0] |= first_bit_mask;
                    for (uint64_t w = w0 + 1; w < ws; ++w) fb[w] = ~0ull;
                    fb[ws] |= below;
                }
            }
            stack.push_back(start);
            node_offset += passed;
            ++restarts;
        }
    }

    // Shrinkable::shrink_single_path (shrinker.rs:178-209); merging the weights = appending the chains
    uint32_t shrink_single_path(uint32_t base_edge) {
        const uint32_t start_node = edge_node[0][base_edge];
        uint32_t mid_node = edge_node[1][base_edge];
        ++calls;
        for (;;) {
            const uint32_t next_edge = node_next[0][mid_node];                 // first_edge(mid_node, Outgoing)
            const uint32_t base_slot = edge_slot[base_edge];
            const uint32_t target = edge_node[1][next_edge];
            uint32_t next_slot;
            if (base_edge < next_edge) {                                       // the higher index first
                next_slot = edge_slot[next_edge];
                remove_edge(next_edge);
                remove_edge(base_edge);
            } else if (base_edge == next_edge) {
                return base_edge;
            } else {
                remove_edge(base_edge);
                next_slot = edge_slot[next_edge];
                remove_edge(next_edge);
            }
            chain_next[chain_last[base_slot]] = next_slot;                     // base_edge_weight.0.merge(next_edge_weight)
            chain_last[base_slot] = chain_last[next_slot];
            ++merges;
            base_edge = add_edge(start_node, target, base_slot);
            mid_node = target;
            if (!degree_is_one(mid_node, 1) || !degree_is_one(mid_node, 0) || mid_node == start_node) return base_edge;
        }
    }

    // Shrinkable::shrink (shrinker.rs:165-176).  Afterwards: edges [0, n_edges) hold the shrunk graph in petgraph's order with their
    // end points as NEW node indices; kept[i] = the old index of the node that is node i after remove_single_vertices
    // (Graph::retain_nodes: indices descending, remove_node = swap_remove)
    void run(std::vector<uint32_t>& kept) {
        for (uint32_t base; (base = next()) != END;) shrink_single_path(base);
        std::vector<uint32_t>& occ = kept;
        occ.resize(n_nodes);
        for (uint32_t i = 0; i < n_nodes; ++i) occ[i] = i;
        uint32_t size = n_nodes;
        for (uint32_t i = n_nodes; i-- > 0;) {
            if (!isolated(i)) continue;                                        // (a node moved in from above was looked at before: it stays)
            const uint32_t last = --size;
            if (i != last) occ[i] = occ[last];
        }
        occ.resize(size);
        std::vector<uint32_t> new_id(n_nodes, END);
        for (uint32_t i = 0; i < size; ++i) new_id[occ[i]] = i;
        for (uint32_t e = 0; e < n_edges; ++e) { edge_node[0][e] = new_id[edge_node[0][e]]; edge_node[1][e] = new_id[edge_node[1][e]]; }
    }
};

}  // namespace katome
