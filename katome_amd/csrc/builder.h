// builder.h -- one GPU's share of a build: the state behind `katome_builder` and the internal steps that the C ABI
// (api.hip) and the sharded driver (dist.hip) compose.  Reference path: Build::create (builder.rs:42-54) ->
// add_read_fastaq (pt_graph.rs:277-315) -> add_single_edge_fastaq (172-198) -> post-pass (333-345).
#pragma once
#include <vector>

#include "common.h"

using namespace katome;

struct katome_builder {
    katome_settings s;
    Profiler prof;
    uint32_t nw = 1;
    bool rc = false;
    Table table;
    bool table_ready = false;
    // tiled counting: (k+span-1)-mers counted first, expanded into `table` before the edges are read out
    Table tiles;
    bool tiles_ready = false;
    uint32_t span = 1;
    // big tiles (span > 16) are first broken into mid tiles of span2 windows (a second tile table), those into k-mers
    Table tiles2;
    bool tiles2_ready = false;
    uint32_t span2 = 0;
    uint64_t stat_tiles2 = 0, stat_tile2_slots = 0;
    // bookkeeping for katome_builder_counts: distinct tiles, tile-table slots, distinct stored k-mers, k-mer-table slots
    // first-seen-order mode (KATOME_FLAG_FIRST_SEEN_ORDER)
    bool first_seen = false;
    uint64_t reads_inserted = 0;       // reads whose records have been handed to an insert so far
    uint32_t seen_read_len = 0;        // read length of the last extraction (records per read follow from it)
    // reads whose windows are not a whole number of tiles: the trailing windows come as plain k-mer records right after
    // the batch's tiles (katome_dev_extract_remainder); first-seen order needs to know where they sit in their reads
    bool rem_pending = false;
    uint32_t rem_win0 = 0, rem_per_read = 0;
    uint64_t last_batch_read0 = 0, last_batch_reads = 0;
    const uint64_t* var_prefix = nullptr;   // variable-length reads: window prefix of the last extraction (device), its reads
    uint64_t var_reads = 0, var_windows = 0;   // and windows; var_seq_base: sequence numbers handed out by earlier batches
    uint64_t var_seq_base = 0;
    const uint64_t* var_rec_prefix = nullptr;  // records before each read for the records just extracted (tiles / left-over windows)
    uint64_t var_records = 0;                  // how many of them: the insert that follows must take exactly these
    uint32_t var_mode = 0, var_span = 1;       // 0 every window, 1 whole tiles, 2 the windows after the last whole tile
    DevBuf edge_seq;                   // sequence number of each edge's first insertion, aligned with edge_key
    // by-packed-key builds: the windows left over after a batch's tiles wait here (records of weight 1), so that the last level can
    // be counted by sorting (table.hip, records_to_edges_sorted) together with the tiles' k-mers; any other consumer of the k-mer
    // table flushes them into it first (flush_rest)
    // by-packed-key builds with two-word tiles and one-word k-mers: the tiles themselves are kept as records too and counted by
    // sorting when the edges are asked for (api.hip, count_tiles_sorted) -- no tile table, no device-scope atomics at any level
    DevBuf tile_recs, tile_recs_count;  // tile_recs_count: device cursor -- the valid ones among them (a skipped read's tiles are all-ones)
    uint64_t tile_recs_n = 0, tile_recs_cap = 0;      // tile_recs_n: records handed over (an upper bound of the cursor)
    bool tile_recs_closed = false;
    bool tile_recs_exact = false;       // every record handed over so far was valid (tile_recs_n IS the count: katome_dev_count_tiles writes in place)
    DevBuf tile_scratch;                // katome_dev_count_tiles: a batch's records when they cannot be made where they are kept
    DevBuf rest_k, rest_count;          // rest_count: device cursor -- how many of them are valid records (reads with N leave invalid ones)
    uint64_t rest_n = 0, rest_cap = 0;  // rest_n: records handed over so far (an upper bound of the cursor)
    bool rest_closed = false;           // too many to keep aside: from now on they go into the table directly
    uint64_t direct_edges = 0;         // BFCounter input: the edges were listed one per line and strand (no table); their count
    uint32_t prune_weight = 0;      // Clean::remove_weak_edges threshold applied when the edges are read out
    uint64_t stat_tiles = 0, stat_tile_slots = 0, stat_kmers = 0, stat_kmer_slots = 0;
    // sorted distinct oriented edges
    bool edges_ready = false;
    DevBuf edge_key, edge_weight;
    uint64_t n_edges = 0;
    // scratch for ordering a batch by table region before it is inserted
    DevBuf scratch_k[2], scratch_w[2];
    // finalized graph
    DevBuf edge_src, edge_dst, edge_label, node_key;
    ShrinkOutput shrunk;               // result of katome_dev_shrink
    DevBuf edge_age;                   // first-seen-order graphs once remove_* has moved edges (PruneGraph::edge_age)
    uint64_t n_nodes = 0;
    bool finalized = false;
};


// ---- internal steps shared with dist.hip (defined in api.hip) --------------------------------------------------------
// find-or-insert `n` records (weights: nullptr = 1 each) into `table`, growing it under the load policy; `origin` places
// the records in the read-ordered stream when the builder tracks first-seen order (its rec0 is set per launch)
int builder_insert(katome_builder* b, Table& table, bool& ready, uint32_t nw, uint64_t hint, const uint64_t* d_records,
                   const uint32_t* d_weights, uint64_t n, SeenOrigin* origin, int phase, hipStream_t stream);
// big tiles -> mid tiles (when the span is large); leaves the tiles that hold k-mers directly in `*last`
int expand_to_last_level(katome_builder* b, Table** last, uint32_t* last_span, hipStream_t stream);
int expand_tiles(katome_builder* b, hipStream_t stream);       // every distinct tile adds its count to its k-mers (b->table); the tile tables go
int flush_rest(katome_builder* b, hipStream_t stream);
int keep_tile_recs(katome_builder* b, const uint64_t* d_records, uint64_t n, uint32_t nwt, bool* kept, hipStream_t stream);   // a batch's valid tile records behind those kept so far
int tile_recs_valid(katome_builder* b, uint64_t* n, hipStream_t stream);      // how many of them there are
// ... counted level by level by sorting, down to the (k-mer, count) records of the last tile level (api.hip)
int tile_recs_to_kmer_records(katome_builder* b, DevBuf& keys, DevBuf& weights, uint64_t* n_records, uint64_t extra_room, hipStream_t stream,
                              DevBuf* first_counts = nullptr);
int sorted_count_mode();            // KATOME_SORTED_COUNT
int sorted_tiles_mode();            // KATOME_SORTED_TILES
bool tile_recs_shape(uint32_t nwt, uint32_t nw, bool first_seen);   // tile / k-mer word counts whose levels are all counted by sorting
bool sorted_fail(const char* level);     // KATOME_SORTED_FAIL (tests)
int flush_tile_recs(katome_builder* b, hipStream_t stream);    // the tile records kept aside -> b->tiles         // the left-over windows kept aside -> b->table
uint32_t mid_span(uint32_t span);      // span of the mid tiles a big tile is broken into (0: expanded directly)
