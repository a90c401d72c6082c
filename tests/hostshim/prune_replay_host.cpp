// Host build of katome_amd/csrc/prune_replay.h for the CPU tests (tests/test_prune_replay_host.py).
#include <string.h>

#include "../../katome_amd/csrc/prune_replay.h"

using namespace katome;

extern "C" {

// out_counts: [n_victims, n_moves, n_new, from_duplicates]; the out arrays must hold `marks` (victims) and `u` (moves) entries
void hs_replay_edges(const uint32_t* pos, const uint32_t* mult, uint64_t u, uint64_t n_edges, uint64_t marks,
                     uint32_t* victims, uint32_t* move_to, uint32_t* move_from, uint64_t* out_counts) {
    EdgeReplay r;
    replay_edges(pos, mult, u, n_edges, marks, r);
    memcpy(victims, r.victims.data(), r.victims.size() * 4);
    memcpy(move_to, r.move_to.data(), r.move_to.size() * 4);
    memcpy(move_from, r.move_from.data(), r.move_from.size() * 4);
    out_counts[0] = r.victims.size(); out_counts[1] = r.move_to.size(); out_counts[2] = r.n_new; out_counts[3] = r.from_duplicates;
}

// out_counts: [n_moves, n_new]; the out arrays must hold 2*m entries
void hs_replay_nodes(const uint32_t* die, uint64_t m, uint64_t n_nodes, uint32_t* move_to, uint32_t* move_from, uint64_t* out_counts) {
    NodeReplay r;
    replay_nodes(die, m, n_nodes, r);
    memcpy(move_to, r.move_to.data(), r.move_to.size() * 4);
    memcpy(move_from, r.move_from.data(), r.move_from.size() * 4);
    out_counts[0] = r.move_to.size(); out_counts[1] = r.n_new;
}

}
