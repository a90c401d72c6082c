// Host-only driver of katome_amd/csrc/prune_replay.h for sanitizer runs (tests/test_sanitizers.py): random passes
// against a literal Vec::swap_remove simulation.  Prints "ok <cases>" or aborts.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <vector>

#include "../../katome_amd/csrc/prune_replay.h"

using namespace katome;

int main() {
    std::mt19937_64 rng(12345);
    int cases = 0;
    for (int it = 0; it < 300; ++it) {
        const uint64_t n = 1 + rng() % 500;
        std::vector<uint32_t> pos, mult;
        uint64_t marks = 0;
        for (uint64_t i = 0; i < n; ++i) if (rng() % 3 == 0) { pos.push_back((uint32_t)i); mult.push_back(1 + (rng() % 5 == 0 ? rng() % 4 : 0)); marks += mult.back(); }
        EdgeReplay er;
        replay_edges(pos.data(), mult.data(), pos.size(), n, marks, er);
        std::vector<uint32_t> arr(n), todo;
        for (uint64_t i = 0; i < n; ++i) arr[i] = (uint32_t)i;
        for (size_t j = 0; j < pos.size(); ++j) for (uint32_t r = 0; r < mult[j]; ++r) todo.push_back(pos[j]);
        std::sort(todo.rbegin(), todo.rend());
        std::vector<uint32_t> victims;
        for (uint32_t d : todo) if (d < arr.size()) { victims.push_back(arr[d]); arr[d] = arr.back(); arr.pop_back(); }
        if (victims != std::vector<uint32_t>(er.victims.p, er.victims.p + er.victims.n) || arr.size() != er.n_new) { fprintf(stderr, "edge replay mismatch\n"); return 1; }
        for (size_t j = 0; j < er.move_to.size(); ++j) if (arr[er.move_to[j]] != er.move_from[j]) { fprintf(stderr, "edge move mismatch\n"); return 1; }
        // nodes: a random subset dies, in random pairs
        std::vector<uint32_t> dying;
        for (uint64_t i = 0; i < n; ++i) if (rng() % 2) dying.push_back((uint32_t)i);
        std::shuffle(dying.begin(), dying.end(), rng);
        std::vector<uint32_t> die;
        while (!dying.empty()) {
            uint32_t a = REPLAY_NONE, b = REPLAY_NONE;
            const int kind = (int)(rng() % 4);
            if (kind == 0 && dying.size() >= 2) { a = dying.back(); dying.pop_back(); b = dying.back(); dying.pop_back(); }
            else if (kind == 1) { a = dying.back(); dying.pop_back(); }
            else if (kind == 2) { b = dying.back(); dying.pop_back(); }
            die.push_back(a); die.push_back(b);
        }
        NodeReplay nr;
        replay_nodes(die.data(), die.size() / 2, n, nr);
        std::vector<uint32_t> nodes(n), where(n);
        for (uint64_t i = 0; i < n; ++i) nodes[i] = where[i] = (uint32_t)i;
        auto remove = [&](uint32_t v) { const uint32_t p = where[v], last = nodes.back(); nodes.pop_back(); if (p < nodes.size()) { nodes[p] = last; where[last] = p; } };
        for (size_t t = 0; t < die.size() / 2; ++t) {
            const uint32_t a = die[2 * t], b = die[2 * t + 1];
            if (a != REPLAY_NONE && b != REPLAY_NONE) { if (where[a] < where[b]) { remove(b); remove(a); } else { remove(a); remove(b); } }
            else if (a != REPLAY_NONE) remove(a);
            else if (b != REPLAY_NONE) remove(b);
        }
        if (nodes.size() != nr.n_new) { fprintf(stderr, "node replay count mismatch\n"); return 1; }
        for (size_t j = 0; j < nr.move_to.size(); ++j) if (nodes[nr.move_to[j]] != nr.move_from[j]) { fprintf(stderr, "node move mismatch\n"); return 1; }
        ++cases;
    }
    printf("ok %d\n", cases);
    return 0;
}
