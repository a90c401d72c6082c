// C entry to katome_amd/csrc/shrink_exact.h for tests/test_shrink_exact_host.py (no GPU)
#include <string.h>

#include "../../katome_amd/csrc/shrink_exact.h"

extern "C" int hs_shrink_exact(const uint32_t* src, const uint32_t* dst, const uint32_t* by_age, uint32_t E, uint32_t N, uint32_t* out_src,
                               uint32_t* out_dst, uint32_t* out_slot, uint32_t* out_kept, uint32_t* chain_next, uint64_t* counts) {
    katome::ShrinkExact s;
    s.init(src, dst, by_age, E, N);
    std::vector<uint32_t> kept;
    s.run(kept);
    for (uint32_t e = 0; e < s.n_edges; ++e) { out_src[e] = s.edge_node[0][e]; out_dst[e] = s.edge_node[1][e]; out_slot[e] = s.edge_slot[e]; }
    if (!kept.empty()) memcpy(out_kept, kept.data(), kept.size() * 4);
    if (E) memcpy(chain_next, s.chain_next.data(), (size_t)E * 4);
    counts[0] = s.n_edges; counts[1] = kept.size(); counts[2] = s.calls; counts[3] = s.merges; counts[4] = s.restarts;
    return 0;
}
