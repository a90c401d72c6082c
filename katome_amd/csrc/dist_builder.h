// dist_builder.h -- state of one rank of the sharded build (dist.hip; the distributed pruning in dist_prune.hip works on it)
#pragma once
#include <algorithm>
#include <vector>

#include "builder.h"
#include "comm.h"

enum XPhase { X_RECORDS, X_KMERS, X_TARGETS, X_IDS, X_RANK_NODES, X_RANK_EDGES, X_GATHER, X_MID_TILES, X_PRUNE, X_COUNT };
extern const char* const XPHASE_NAMES[X_COUNT];

struct katome_dist_builder {
    katome_settings s;
    katome_comm* comm = nullptr;
    katome_builder* b = nullptr;             // this rank's single-GPU builder (tables, sorted edges)
    uint32_t nw = 1;
    bool rc = false, first_seen = false;
    // the plan, the same on every rank (a function of k and the read length)
    bool planned = false;
    uint32_t read_len = 0, W = 0, span = 1, tiles_per_read = 0, rest = 0, nwt = 1;
    uint64_t reads_end = 0;                  // one past the last read this rank has added (first-seen: bounds the sequence numbers)
    bool finalized = false;
    // Which records travel.  Few ranks share few links: every rank counts its own reads down to k-mers and
    // sends each DISTINCT k-mer once ("local first": one exchange, 12 B per k-mer and rank).  Many ranks: tiles, mid tiles and
    // k-mer records are routed to owners level by level (no level is counted twice, at the price of three exchanges).
    bool local_first = false;
    // Round 4 (KATOME_DIST_ROUTE=supermers; by packed key, k <= 31, reads of one length): ONE exchange before anything is counted --
    // a read travels as its supermers (supermer.hip: runs of windows with one minimizer, a 16-byte record each, owner = a hash of the
    // minimizer), every rank then counts what it received as one GPU counts its own reads.  `owner_m` > 0: a k-mer's / node's owner
    // is named by the minimizer of that many bases of its core (the finalize's target look-ups use the same function).
    bool want_supermers = false, supermers = false;
    uint32_t owner_m = 0, sm_slots = 0;
    DevBuf sm_recs;                          // [sm_n][2]: per add_reads call the reads' slots, then that call's spill region
    uint64_t sm_n = 0, sm_cap = 0;
    hipStream_t xstream = nullptr;           // the exchanges of route_weighted run here, beside the insertions on the build's stream
    // this rank's share of the numbered graph
    DevBuf edge_src, edge_dst, edge_label, node_key, edge_gid, node_gid;
    uint64_t n_edges = 0, n_nodes = 0, total_edges = 0, total_nodes = 0, node_base = 0;
    katome::ExchangeStats xstats[X_COUNT];
    // kept for the stages that run on the sharded graph (dist_prune.hip; first-seen order only): every edge's source as the
    // LOCAL index of the node (all out-edges of a node live on the node's owner) and its target as (owner rank, local index there)
    DevBuf edge_lsrc, edge_drank, edge_dlocal;
    DevBuf edge_age;                         // after katome_dist_remove_dead_paths: first-seen index each surviving edge had (u64)
    uint64_t n_src = 0;                      // this rank's nodes [0, n_src) have out-edges (ascending by key), the rest do not
    bool dead_paths_removed = false;         // katome_dist_remove_dead_paths has run to its fixpoint on this sharded graph
    bool gathered = false;                   // katome_dist_gather has consumed the ranks' shares

    int world() const { return comm->world(); }
    int rank() const { return comm->rank(); }
    // all-to-all of records grouped by destination, accounted to `phase`
    // (RCCL only enqueues: with the builder's profile switched on the exchange is timed with HIP events on its stream)
    struct XEvent { int phase; hipEvent_t a, b; };
    std::vector<XEvent> xevents;
    int xchg(int phase, const void* send, const uint64_t* send_cnt, void* recv, const uint64_t* recv_cnt, size_t elem_bytes, hipStream_t stream,
             bool one_round = false, uint64_t known_max = katome_comm::MAX_UNKNOWN, const uint64_t* send_off = nullptr) {
        const katome::ExchangeStats before = comm->stats;
        hipEvent_t ea = nullptr, eb = nullptr;
        bool timed = b->prof.on && hipEventCreate(&ea) == hipSuccess;
        if (timed && hipEventCreate(&eb) != hipSuccess) { (void)hipEventDestroy(ea); ea = nullptr; timed = false; }
        // (the sharded pruning makes dozens of exchanges per pass: finished pairs are folded into the phase's time instead of piling up)
        if (timed && xevents.size() >= 256) fold_xevents();
        if (timed) (void)hipEventRecord(ea, stream);
        const int rc = comm->exchange(send, send_cnt, recv, recv_cnt, elem_bytes, 1, stream, one_round, known_max, send_off);
        if (timed) { (void)hipEventRecord(eb, stream); xevents.push_back({phase, ea, eb}); }
        if (rc != KATOME_OK) return rc;
        katome::ExchangeStats& x = xstats[phase];
        x.calls += comm->stats.calls - before.calls; x.bytes_out += comm->stats.bytes_out - before.bytes_out;
        x.bytes_in += comm->stats.bytes_in - before.bytes_in;
        if (!timed) x.ms += comm->stats.ms - before.ms;
        for (int p = 0; p < world(); ++p) if (p != rank()) x.max_pair_bytes = std::max<uint64_t>(x.max_pair_bytes, send_cnt[p] * elem_bytes);
        return KATOME_OK;
    }
    // the pairs whose second event has completed: their time goes to their phase, the events are destroyed
    void fold_xevents() {
        size_t kept = 0;
        for (auto& e : xevents) {
            float ms = 0;
            if (hipEventQuery(e.b) == hipSuccess) {
                if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) xstats[e.phase].ms += ms;
                (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
            } else xevents[kept++] = e;
        }
        xevents.resize(kept);
    }
    ~katome_dist_builder() {
        for (auto& e : xevents) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        if (xstream) { (void)hipSetDevice(s.device); dev_retire_stream(xstream); (void)hipStreamDestroy(xstream); }
    }
};

