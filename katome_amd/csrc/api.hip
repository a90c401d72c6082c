// api.hip -- the C ABI of include/katome_gpu.h: the build driver behind `Build::create`
// (reference src/katome/algorithms/builder.rs:42-54) and the PtGraph::create post-pass
// (collections/graphs/pt_graph.rs:333-345), composed from the kernels in extract.hip, table.hip
// and radix.hip.  No CPU fallback: every entry that needs the device fails with KATOME_E_DEVICE
// when there is none.
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <string>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <sys/mman.h>

#include "builder.h"
#include "comm.h"

extern "C" {

uint32_t katome_abi_version(void) { return KATOME_ABI_VERSION; }
const char* katome_last_error(void) { return get_error(); }
uint32_t katome_record_words(uint32_t k) { return (uint32_t)key_words_for_k(k); }

int katome_builder_create(const katome_settings* s, katome_builder** out) {
    if (!s || !out) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    KCHECK(check_k(s->k));
    KCHECK(use_device(s->device));
    katome_builder* b = new (std::nothrow) katome_builder();
    if (!b) { set_error("out of host memory"); return KATOME_E_OOM; }
    b->s = *s;
    b->nw = (uint32_t)key_words_for_k(s->k);
    b->rc = s->reverse_complement != 0;
    b->first_seen = (s->flags & KATOME_FLAG_FIRST_SEEN_ORDER) != 0;
    if ((s->flags & KATOME_FLAG_REMOVE_DEAD_PATHS) && !b->first_seen) {
        delete b;
        set_error("KATOME_FLAG_REMOVE_DEAD_PATHS needs KATOME_FLAG_FIRST_SEEN_ORDER: the reference's pruning depends on petgraph's numbering");
        return KATOME_E_ARG;
    }
    b->table.track_seen = b->tiles.track_seen = b->tiles2.track_seen = b->first_seen;
    *out = b;
    return KATOME_OK;
}

void katome_builder_destroy(katome_builder* b) {
    if (!b) return;
    (void)hipSetDevice(b->s.device);
    delete b;
}

// first-seen order numbers window i of read r as r * 2W + i: every fixed-length batch of a build must have the same length
// (reads of several lengths go through katome_dev_extract_var*, which numbers by window prefix sums)
static int check_seen_len(katome_builder* b, uint32_t read_len) {
    if (b->first_seen && b->reads_inserted && b->seen_read_len && b->seen_read_len != read_len) {
        set_error("first-seen order: fixed-length batches of %u and %u bases in one build (use the variable-length entry points)", b->seen_read_len, read_len);
        return KATOME_E_ARG;
    }
    return KATOME_OK;
}

int katome_dev_extract_fixed(katome_builder* b, const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len,
                             const uint8_t* d_skip, uint64_t* d_records, void* stream) {
    KCHECK_HIP(hipSetDevice(b->s.device));
    KCHECK(check_seen_len(b, read_len));
    PhaseScope ps(b->prof, PH_EXTRACT, (hipStream_t)stream);
    b->seen_read_len = read_len;
    return launch_extract_fixed(b->s.k, b->rc, d_packed, n_reads, read_len, d_skip, d_records, (hipStream_t)stream, 1,
                                b->first_seen && b->rc);
}

// Tiled counting (table.hip): largest span in 2..32 that divides the windows per read and keeps the tile in 128 bits
uint32_t katome_tile_span(uint32_t k, uint32_t read_len) {
    if (getenv("KATOME_NO_TILES") || read_len < k) return 1;
    const uint32_t W = read_len - k + 1;
    if (const char* e = getenv("KATOME_TILE_SPAN")) {        // experiments: any span that divides W and fits 128 bits
        const uint32_t s = (uint32_t)atoi(e);
        if (s >= 1 && W % s == 0 && k + s - 1 <= 63) return s;
    }
    for (uint32_t s = 32; s >= 2; --s) if (W % s == 0 && k + s - 1 <= 63) return s;
    return 1;
}
// how a read of `read_len` bases is best counted: `tiles` tiles of `span` windows from the front, then `remainder`
// single windows.  Fewest table insertions per read; spans above 16 without a divisor to break them into mid tiles
// (two-level expansion) are charged for it.  Tiles may take up to `max_tile_words` u64 words (3: 95 bases, which is what
// lets k = 63 be tiled at all).  Returns 0 (span 1) when tiling does not pay or is switched off.
uint32_t katome_tile_plan_limited(uint32_t k, uint32_t read_len, uint32_t max_tile_words, uint32_t* span, uint32_t* tiles,
                                  uint32_t* remainder) {
    uint32_t best_s = 1, best_cost = 0xFFFFFFFFu;
    const uint32_t W = read_len >= k ? read_len - k + 1 : 0;
    const uint32_t max_bases = max_tile_words >= 3 ? 95u : max_tile_words == 2 ? 63u : 31u;
    if (W >= 2 && !getenv("KATOME_NO_TILES")) {
        if (const char* e = getenv("KATOME_TILE_SPAN")) {
            const uint32_t s = (uint32_t)atoi(e);
            if (s >= 2 && s <= W && k + s - 1 <= max_bases) { best_s = s; best_cost = 0; }
        }
        for (uint32_t s = 2; best_cost != 0 && s <= 33 && s <= W && k + s - 1 <= max_bases; ++s) {
            bool breakable = s <= 16;
            for (uint32_t d = 3; d <= 8 && !breakable; ++d) breakable = s % d == 0;
            const uint32_t cost = W / s + W % s + (breakable ? 0 : 4);
            if (cost < best_cost || (cost == best_cost && s > best_s)) { best_cost = cost; best_s = s; }
        }
        if (best_cost != 0 && best_cost >= W) best_s = 1;
    }
    if (span) *span = best_s;
    if (tiles) *tiles = best_s > 1 ? W / best_s : 0;
    if (remainder) *remainder = best_s > 1 ? W % best_s : W;
    return best_s > 1;
}
uint32_t katome_tile_plan(uint32_t k, uint32_t read_len, uint32_t* span, uint32_t* tiles, uint32_t* remainder) {
    return katome_tile_plan_limited(k, read_len, 3, span, tiles, remainder);
}
uint32_t katome_tile_words(uint32_t k, uint32_t span) { return (uint32_t)key_words_for_k(k + span - 1); }

int katome_dev_extract_tiles(katome_builder* b, const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len, uint32_t span,
                             const uint8_t* d_skip, uint64_t* d_records, void* stream) {
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (span < 1 || b->s.k + span - 1 > 95) { set_error("bad tile span %u", span); return KATOME_E_ARG; }
    KCHECK(check_seen_len(b, read_len));
    PhaseScope ps(b->prof, PH_EXTRACT, (hipStream_t)stream);
    b->seen_read_len = read_len;
    return launch_extract_fixed(b->s.k, b->rc, d_packed, n_reads, read_len, d_skip, d_records, (hipStream_t)stream, span,
                                b->first_seen && b->rc);
}

int katome_dev_extract_remainder(katome_builder* b, const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len, uint32_t span,
                                 const uint8_t* d_skip, uint64_t* d_records, void* stream) {
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (span < 1 || read_len < b->s.k) { set_error("bad tile span %u", span); return KATOME_E_ARG; }
    const uint32_t W = read_len - b->s.k + 1, first = (W / span) * span, rest = W - first;
    if (rest == 0) return KATOME_OK;
    KCHECK(check_seen_len(b, read_len));
    PhaseScope ps(b->prof, PH_EXTRACT, (hipStream_t)stream);
    b->seen_read_len = read_len;
    b->rem_pending = true; b->rem_win0 = first; b->rem_per_read = rest;
    return launch_extract_fixed(b->s.k, b->rc, d_packed, n_reads, read_len, d_skip, d_records, (hipStream_t)stream, 1,
                                b->first_seen && b->rc, first, rest);
}

static int extract_var_common(katome_builder* b, const uint8_t* d_packed, uint64_t packed_bytes, const uint64_t* d_byte_off,
                              const uint32_t* d_len, const uint64_t* d_rec_prefix, const uint64_t* d_win_prefix, uint64_t n_reads,
                              uint64_t n_records, uint64_t total_windows, uint32_t span, uint32_t mode, uint64_t* d_records, void* stream) {
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (mode == 1 && (span < 2 || b->s.k + span - 1 > 95)) { set_error("bad tile span %u", span); return KATOME_E_ARG; }
    if (b->first_seen) {
        if (b->reads_inserted) { set_error("first-seen order: fixed- and variable-length batches cannot be mixed in one build"); return KATOME_E_UNSUPPORTED; }
        // the insert that follows reads these
        b->var_prefix = d_win_prefix; b->var_reads = n_reads; b->var_windows = total_windows;
        b->var_rec_prefix = d_rec_prefix; b->var_records = n_records; b->var_mode = mode; b->var_span = span;
    }
    PhaseScope ps(b->prof, PH_EXTRACT, (hipStream_t)stream);
    return launch_extract_var(b->s.k, b->rc, d_packed, packed_bytes, d_byte_off, d_len, d_rec_prefix, n_reads, n_records,
                              d_records, (hipStream_t)stream, b->first_seen && b->rc, span, mode);
}

int katome_dev_extract_var(katome_builder* b, const uint8_t* d_packed, uint64_t packed_bytes, const uint64_t* d_byte_off,
                           const uint32_t* d_len, const uint64_t* d_win_prefix, uint64_t n_reads, uint64_t total_windows,
                           uint64_t* d_records, void* stream) {
    return extract_var_common(b, d_packed, packed_bytes, d_byte_off, d_len, d_win_prefix, d_win_prefix, n_reads, total_windows,
                              total_windows, 1, 0, d_records, stream);
}
int katome_dev_extract_var_tiles(katome_builder* b, const uint8_t* d_packed, uint64_t packed_bytes, const uint64_t* d_byte_off,
                                 const uint32_t* d_len, const uint64_t* d_tile_prefix, const uint64_t* d_win_prefix, uint64_t n_reads,
                                 uint64_t total_tiles, uint64_t total_windows, uint32_t span, uint64_t* d_records, void* stream) {
    return extract_var_common(b, d_packed, packed_bytes, d_byte_off, d_len, d_tile_prefix, d_win_prefix, n_reads, total_tiles,
                              total_windows, span, 1, d_records, stream);
}
int katome_dev_extract_var_remainder(katome_builder* b, const uint8_t* d_packed, uint64_t packed_bytes, const uint64_t* d_byte_off,
                                     const uint32_t* d_len, const uint64_t* d_rest_prefix, const uint64_t* d_win_prefix, uint64_t n_reads,
                                     uint64_t total_rest, uint64_t total_windows, uint32_t span, uint64_t* d_records, void* stream) {
    if (span < 1) { set_error("bad tile span %u", span); return KATOME_E_ARG; }
    return extract_var_common(b, d_packed, packed_bytes, d_byte_off, d_len, d_rest_prefix, d_win_prefix, n_reads, total_rest,
                              total_windows, span, 2, d_records, stream);
}

int katome_dev_partition(int device, const uint64_t* d_records, const uint32_t* d_values, uint64_t n_records, uint32_t key_words,
                         uint32_t n_parts, uint64_t* d_out, uint32_t* d_values_out, uint64_t* h_counts, void* stream) {
    KCHECK(use_device(device));
    return dev_partition(d_records, d_values, n_records, key_words, n_parts, d_out, d_values_out, h_counts, (hipStream_t)stream);
}
int katome_dev_partition_core(int device, const uint64_t* d_records, const uint32_t* d_values, uint64_t n_records, uint32_t key_words,
                              uint32_t core_shift, uint32_t core_bases, uint32_t n_parts, uint64_t* d_out, uint32_t* d_values_out,
                              uint64_t* h_counts, void* stream) {
    KCHECK(use_device(device));
    if (core_bases == 0) { set_error("partition_core: core_bases must be > 0"); return KATOME_E_ARG; }
    return dev_partition(d_records, d_values, n_records, key_words, n_parts, d_out, d_values_out, h_counts, (hipStream_t)stream,
                         core_shift, core_bases);
}
uint32_t katome_key_owner(const uint64_t* key, uint32_t key_words, uint32_t core_shift, uint32_t core_bases, uint32_t n_parts) {
    if (key_words == 1) {
        Key<1> a; a.w[0] = key[0];
        return (uint32_t)(core_bases ? core_owner(a, core_shift, core_bases, n_parts) : whole_key_owner(a, n_parts));
    }
    if (key_words == 3) {
        Key<3> a; a.w[0] = key[0]; a.w[1] = key[1]; a.w[2] = key[2];
        return (uint32_t)(core_bases ? core_owner(a, core_shift, core_bases, n_parts) : whole_key_owner(a, n_parts));
    }
    Key<2> a; a.w[0] = key[0]; a.w[1] = key[1];
    return (uint32_t)(core_bases ? core_owner(a, core_shift, core_bases, n_parts) : whole_key_owner(a, n_parts));
}

}  // extern "C"

// Table capacity policy.  A chunk of records may only be handed to the insert kernel when even in the
// worst case (every record a new key) the table stays below MAX_LOAD, which bounds every probe sequence;
// the table is doubled when its real occupancy passes GROW_LOAD.  `room` returns how many records may go in now.
static constexpr double MAX_LOAD = 0.9, GROW_LOAD = 0.6;

static int table_budget(katome_builder* b, double frac, uint64_t* slots) {
    size_t free_b = 0, total_b = 0;
    KCHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += dev_cached_bytes();           // cached blocks are handed back when an allocation needs them
    *slots = (uint64_t)(free_b * frac) / (b->nw == 1 ? 16 : 32);
    return KATOME_OK;
}

static int ensure_table(katome_builder* b, Table& table, bool& ready, uint32_t nw, uint64_t hint, uint64_t incoming,
                        uint64_t* room, hipStream_t stream) {
    if (!ready) {
        uint64_t want = hint ? hint : std::min<uint64_t>(incoming, 1ull << 28) * 2;
        want = std::max<uint64_t>(want, 1u << 16);
        uint64_t budget = 0;
        KCHECK(table_budget(b, 0.5, &budget));
        budget = budget * 16 / (nw == 1 ? 16 : 32) * (b->nw == 1 ? 1 : 2);     // table_budget counts b->nw-sized slots
        if (want > budget) want = budget;
        KCHECK(table_alloc(table, nw, want, stream));
        ready = true;
    }
    for (;;) {
        uint64_t occ = 0;
        KCHECK(table_occupied(table, &occ, stream));
        const uint64_t limit = (uint64_t)(MAX_LOAD * (double)table.cap);
        const uint64_t r = limit > occ ? limit - occ : 0;
        const bool crowded = (double)occ > GROW_LOAD * (double)table.cap;
        if (!crowded && r >= std::min<uint64_t>(incoming, 1u << 20)) { *room = r; return KATOME_OK; }
        uint64_t want = table.cap * 2, budget = 0;
        KCHECK(table_budget(b, 0.9, &budget));
        budget = budget * 16 / (nw == 1 ? 16 : 32) * (b->nw == 1 ? 1 : 2);
        if (want > budget) want = budget;
        if (want <= table.cap + table.cap / 8) {
            if (r > 0) { *room = r; return KATOME_OK; }      // cannot grow: run on, up to the hard limit
            set_error("k-mer table is full (%llu keys in %llu slots) and cannot grow in device memory",
                      (unsigned long long)occ, (unsigned long long)table.cap);
            return KATOME_E_OOM;
        }
        KCHECK(table_grow(table, want, stream));
    }
}
static int ensure_table(katome_builder* b, uint64_t incoming, uint64_t* room, hipStream_t stream) {
    return ensure_table(b, b->table, b->table_ready, b->nw, b->s.table_slots_hint, incoming, room, stream);
}

int builder_insert(katome_builder* b, Table& table, bool& ready, uint32_t nw, uint64_t hint, const uint64_t* d_records,
                   const uint32_t* d_weights, uint64_t n, SeenOrigin* origin, int phase, hipStream_t stream) {
    for (uint64_t done = 0; done < n;) {
        uint64_t room = 0;
        KCHECK(ensure_table(b, table, ready, nw, hint, n - done, &room, stream));
        const uint64_t m = std::min(n - done, room);
        PhaseScope ps(b->prof, phase, stream);
        if (origin) origin->rec0 = done;               // (idx / pairs of an explicit origin are indexed from the batch's start)
        KCHECK(table_insert(table, d_records + done * nw, d_weights ? d_weights + done : nullptr, m, stream, table.track_seen ? origin : nullptr));
        done += m;
    }
    return KATOME_OK;
}

// span of the mid tiles a big tile is broken into: the divisor of `span` in 2..8 closest to 6; 0 = expand directly
uint32_t mid_span(uint32_t span) {
    if (span <= 16 || getenv("KATOME_ONE_LEVEL_TILES")) return 0;
    if (const char* e = getenv("KATOME_MID_SPAN")) {         // experiments: any divisor of the span
        const uint32_t s = (uint32_t)atoi(e);
        if (s >= 2 && s < span && span % s == 0) return s;
    }
    static const uint32_t pref[] = {6, 5, 7, 4, 8, 9, 3, 2};       // (9 before 3: k = 40 at 150 bp, span 27 -- 103.5 against 106.3 ms per 50 M reads)
    for (uint32_t s : pref) if (span % s == 0) return s;
    return 0;
}

// walk `from` in slot ranges small enough that even if every sub-window of the range were a new key, `to` stays
// under its load limit (so it keeps the size its hint gave it); every tile adds its count to its n_sub sub-windows
static int expand_level(katome_builder* b, Table& from, Table& to, bool& to_ready, uint32_t to_nw, uint64_t to_hint,
                        uint32_t sub_len, uint32_t n_sub, uint32_t stride, int phase, hipStream_t stream) {
    for (uint64_t s0 = 0; s0 < from.cap;) {
        uint64_t room = 0;
        KCHECK(ensure_table(b, to, to_ready, to_nw, to_hint, (uint64_t)n_sub << 20, &room, stream));
        const uint64_t slots = std::max<uint64_t>(std::min<uint64_t>(from.cap - s0, room / n_sub), 1);
        PhaseScope ps(b->prof, phase, stream);
        KCHECK(table_expand_tiles(from, s0, s0 + slots, to, sub_len, n_sub, stride, b->rc, stream));
        s0 += slots;
    }
    return KATOME_OK;
}

// big tiles -> mid tiles (when the span is large); leaves the tiles that hold k-mers directly in `*last`
int expand_to_last_level(katome_builder* b, Table** last, uint32_t* last_span, hipStream_t stream) {
    if (b->tiles2_ready && b->tiles.cap == 0) { *last = &b->tiles2; *last_span = b->span2; return KATOME_OK; }      // (done before)
    uint64_t n_tiles = 0;
    KCHECK(table_occupied(b->tiles, &n_tiles, stream));
    b->stat_tiles = n_tiles; b->stat_tile_slots = b->tiles.cap;
    b->span2 = mid_span(b->span);
    *last = &b->tiles; *last_span = b->span;
    if (b->span2 && n_tiles) {
        const uint32_t kk2 = b->s.k + b->span2 - 1, n_sub = b->span / b->span2;
        // the mid-tile table is sized from what is known by now: n_tiles distinct tiles make at most n_tiles * n_sub mid
        // tiles (C3: 1.6 per tile); five slots per tile keep its load near 1/3 (78 -> 68 ms for this level at C3)
        const uint64_t mid_hint = std::max<uint64_t>(b->s.table_slots_hint / 4, std::min<uint64_t>(n_tiles * 5, n_tiles * n_sub * 2));
        KCHECK(expand_level(b, b->tiles, b->tiles2, b->tiles2_ready, (uint32_t)key_words_for_k(kk2), mid_hint,
                            kk2, n_sub, b->span2, PH_EXPAND_MID, stream));
        b->tiles.release();
        KCHECK(table_occupied(b->tiles2, &b->stat_tiles2, stream));
        b->stat_tile2_slots = b->tiles2.cap;
        *last = &b->tiles2; *last_span = b->span2;
    }
    return KATOME_OK;
}

// every distinct tile adds its count to its k-mers; afterwards the tile tables are released
int expand_tiles(katome_builder* b, hipStream_t stream) {
    if (b->tile_recs_n) KCHECK(flush_tile_recs(b, stream));
    if (b->rest_n) KCHECK(flush_rest(b, stream));
    if (!b->tiles_ready) return KATOME_OK;
    Table* last = nullptr; uint32_t last_span = 1;
    KCHECK(expand_to_last_level(b, &last, &last_span, stream));
    if (b->stat_tiles) {
        // Without a hint from the caller the k-mer table is sized from what is known by now: the distinct tiles of the last
        // level hold at most (their number x span) distinct k-mers (C3: 36 % of that).  Starting small and doubling eight
        // times re-inserted every k-mer once more on the way (C3: +40 ms).
        uint64_t hint = b->s.table_slots_hint;
        if (!hint) {
            const uint64_t last_tiles = b->span2 ? b->stat_tiles2 : b->stat_tiles;
            hint = std::max<uint64_t>(last_tiles * last_span, 1u << 16);
            if (b->table_ready && b->table.cap < hint) {          // (left-over windows went in first: one growth step instead of many)
                uint64_t budget = 0;
                KCHECK(table_budget(b, 0.9, &budget));
                const uint64_t want = std::min(hint, budget);
                if (want > b->table.cap + b->table.cap / 8) KCHECK(table_grow(b->table, want, stream));
            }
        }
        KCHECK(expand_level(b, *last, b->table, b->table_ready, b->nw, hint, b->s.k, last_span, 1, PH_EXPAND_TILES, stream));
    }
    b->tiles.release();
    b->tiles2.release();
    b->tiles_ready = false; b->tiles2_ready = false;
    return KATOME_OK;
}

// how many 8-bit region passes to run in front of an insert, from the table size (KATOME_REGION_PASSES overrides)
static int region_passes(uint64_t table_bytes) {
    (void)table_bytes;
    if (const char* e = getenv("KATOME_REGION_PASSES")) return std::max(0, std::min(2, atoi(e)));
    return 0;
}

// BFCounter input: one edge per kept line and strand, never merged (add_single_edge_bfc, pt_graph.rs:201-213, calls
// add_edge unconditionally).  d_fwd: the lines' k-mers as packed keys in line order, d_w their weights.  Leaves the
// builder with its sorted edge list, as katome_dev_edges would.
static int bfc_set_edges(katome_builder* b, const uint64_t* d_fwd, const uint32_t* d_w, uint64_t n_lines, hipStream_t stream) {
    const uint64_t E = n_lines * (b->rc ? 2 : 1);
    const uint32_t nw = b->nw;
    if (b->edges_ready || b->table_ready || b->tiles_ready) { set_error("BFCounter input cannot be mixed with counted reads"); return KATOME_E_ARG; }
    b->n_edges = E; b->direct_edges = E;
    KCHECK(b->edge_key.alloc((E + 1) * 8 * nw, stream));
    KCHECK(b->edge_weight.alloc((E + 1) * 4, stream));
    PhaseScope ps(b->prof, PH_SORT_EDGES, stream);
    if (b->first_seen) {
        if (E >= (1ull << 32)) { set_error("first-seen order: more than 2^32 edges on one GPU"); return KATOME_E_UNSUPPORTED; }
        DevBuf raw_w(stream), raw_seq(stream), idx(stream);
        KCHECK(raw_w.alloc((E + 1) * 4)); KCHECK(raw_seq.alloc((E + 1) * 8)); KCHECK(idx.alloc((E + 1) * 4));
        KCHECK(dev_bfc_edges(d_fwd, d_w, n_lines, b->s.k, b->rc, b->edge_key.as<u64>(), raw_w.as<u32>(), raw_seq.as<u64>(), stream));
        KCHECK(dev_iota(idx.as<u32>(), E, stream));
        KCHECK(dev_sort_bufs(b->edge_key, &idx, E, nw, 2 * b->s.k, stream));
        KCHECK(b->edge_seq.alloc((E + 1) * 8, stream));
        KCHECK(dev_gather_u64(raw_seq.as<u64>(), idx.as<u32>(), E, b->edge_seq.as<u64>(), stream));
        KCHECK(dev_gather_u32(raw_w.as<u32>(), idx.as<u32>(), E, b->edge_weight.as<u32>(), stream));
    } else {
        KCHECK(dev_bfc_edges(d_fwd, d_w, n_lines, b->s.k, b->rc, b->edge_key.as<u64>(), b->edge_weight.as<u32>(), nullptr, stream));
        KCHECK(dev_sort_bufs(b->edge_key, &b->edge_weight, E, nw, 2 * b->s.k, stream));   // stable: a k-mer's lines stay in file order
    }
    b->edges_ready = true;
    return KATOME_OK;
}

int sorted_count_mode() {
    static const int mode = getenv("KATOME_SORTED_COUNT") ? atoi(getenv("KATOME_SORTED_COUNT")) : 1;
    return mode;
}

// plain (weight-1 or weighted) records into the k-mer table, by packed key (no origin)
static int insert_plain(katome_builder* b, const uint64_t* d_records, const uint32_t* d_weights, uint64_t n_records, hipStream_t stream) {
    for (uint64_t done = 0; done < n_records;) {
        uint64_t room = 0;
        KCHECK(ensure_table(b, n_records - done, &room, stream));
        const uint64_t n = std::min(n_records - done, room);
        PhaseScope ps(b->prof, PH_INSERT, stream);
        KCHECK(table_insert(b->table, d_records + done * b->nw, d_weights ? d_weights + done : nullptr, n, stream, nullptr));
        done += n;
    }
    return KATOME_OK;
}

// how many valid records wait in b->rest_k (the device cursor; b->rest_n counts what was handed over, skipped reads included)
static int rest_valid(katome_builder* b, uint64_t* n, hipStream_t stream) {
    *n = 0;
    if (!b->rest_n || !b->rest_count.p) return KATOME_OK;
    KCHECK_HIP(hipMemcpyAsync(n, b->rest_count.p, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    return KATOME_OK;
}
static void rest_reset(katome_builder* b) { b->rest_k.release(); b->rest_count.release(); b->rest_n = b->rest_cap = 0; }

int flush_rest(katome_builder* b, hipStream_t stream) {
    uint64_t n_valid = 0;
    KCHECK(rest_valid(b, &n_valid, stream));
    if (n_valid && b->first_seen) {            // tagged records: back into keys + their two sequence numbers for the table
        DevBuf keys(stream), pairs(stream);
        KCHECK(keys.alloc(n_valid * 8 * b->nw + 16)); KCHECK(pairs.alloc(n_valid * 16 + 16));
        KCHECK(table_tagged_to_pairs(b->rest_k.as<u64>(), n_valid, b->nw, 2ull * (b->seen_read_len - b->s.k + 1), keys.as<u64>(), pairs.as<u64>(), stream));
        SeenOrigin origin;
        origin.pairs = pairs.as<u64>(); origin.rc = b->rc;
        KCHECK(builder_insert(b, b->table, b->table_ready, b->nw, b->s.table_slots_hint, keys.as<u64>(), nullptr, n_valid, &origin, PH_INSERT, stream));
    } else if (n_valid) {
        KCHECK(insert_plain(b, b->rest_k.as<u64>(), nullptr, n_valid, stream));
    }
    rest_reset(b);
    b->rest_closed = true;
    return KATOME_OK;
}

// KATOME_SORTED_FAIL=mid|last: that level's counting by sorting reports a group too large -- tests of the way back into the tables
bool sorted_fail(const char* level) {
    static const char* at = getenv("KATOME_SORTED_FAIL");
    return at && !strcmp(at, level);
}
// KATOME_SORTED_TILES: 2 (default) both tile levels are counted by sorting -- the big tiles' records are kept aside batch by batch
// (two-word tiles of one-word k-mers, either numbering; anything else takes the table) --, 1 the big tiles in their table and only
// the mid tiles by sorting, out of that table; 0 both tile levels in tables.  C3 by packed key: 200 / 210 / 239 ms per build.
int sorted_tiles_mode() {
    static const int mode = getenv("KATOME_SORTED_TILES") ? atoi(getenv("KATOME_SORTED_TILES")) : 2;
    return mode;
}

// Tile / k-mer shapes whose levels are all counted by sorting: tiles of two or three words (32..95 bases) over k-mers of one or two
// (round 4: three-word tiles -- every k from 32 to 63 at 150 bp, the reference's example k = 40 and BASELINE's k = 63 -- and two-word
// tiles of two-word k-mers); in the reference's numbering two-word tiles of one-word k-mers (the tagged kernels' shapes).
// KATOME_SORTED_WIDE=0: the round-3 rule (A/B against the tables).
bool tile_recs_shape(uint32_t nwt, uint32_t nw, bool first_seen) {
    static const bool wide = !getenv("KATOME_SORTED_WIDE") || atoi(getenv("KATOME_SORTED_WIDE")) != 0;
    if (first_seen || !wide) return nwt == 2 && nw == 1;
    return (nwt == 2 || nwt == 3) && nw <= 2 && nw <= nwt;
}

// the tile records kept aside (builder.h) go into the tile table after all: another consumer wants the table, or the sorted
// counting of the tiles gave up
int tile_recs_valid(katome_builder* b, uint64_t* n, hipStream_t stream) {
    *n = 0;
    if (b->tile_recs_exact) { *n = b->tile_recs_n; return KATOME_OK; }       // (every record kept so far was valid: nothing to ask the device)
    if (!b->tile_recs_n || !b->tile_recs_count.p) return KATOME_OK;
    KCHECK_HIP(hipMemcpyAsync(n, b->tile_recs_count.p, 8, hipMemcpyDeviceToHost, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    return KATOME_OK;
}
int flush_tile_recs(katome_builder* b, hipStream_t stream) {
    if (b->tile_recs_n) {
        const uint32_t nwt = (uint32_t)key_words_for_k(b->s.k + b->span - 1);
        uint64_t n = 0;
        KCHECK(tile_recs_valid(b, &n, stream));
        b->tile_recs_n = 0;
        if (b->first_seen && n) {            // tagged records: back into keys + their two sequence numbers for the table
            DevBuf keys(stream), pairs(stream);
            KCHECK(keys.alloc(n * 8 * nwt + 16)); KCHECK(pairs.alloc(n * 16 + 16));
            KCHECK(table_tagged_to_pairs(b->tile_recs.as<u64>(), n, nwt, 2ull * (b->seen_read_len - b->s.k + 1), keys.as<u64>(), pairs.as<u64>(), stream));
            b->tile_recs.release();
            SeenOrigin origin;
            origin.pairs = pairs.as<u64>(); origin.rc = b->rc;
            KCHECK(builder_insert(b, b->tiles, b->tiles_ready, nwt, b->s.table_slots_hint / 4, keys.as<u64>(), nullptr, n, &origin, PH_INSERT_TILES, stream));
        } else
        for (uint64_t done = 0; done < n;) {
            uint64_t room = 0;
            KCHECK(ensure_table(b, b->tiles, b->tiles_ready, nwt, b->s.table_slots_hint / 4, n - done, &room, stream));
            const uint64_t m = std::min(n - done, room);
            PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
            KCHECK(table_insert(b->tiles, b->tile_recs.as<u64>() + done * nwt, nullptr, m, stream, nullptr));
            done += m;
        }
    }
    b->tile_recs.release(); b->tile_recs_count.release();
    b->tile_recs_n = b->tile_recs_cap = 0;
    b->tile_recs_exact = false;
    b->tile_recs_closed = true;
    return KATOME_OK;
}

// room for n more tile records behind those kept aside; *ok = false: no room (or over the limit) -- what was kept has gone into the
// tile table and later batches follow it there
static int reserve_tile_recs(katome_builder* b, uint64_t n, uint32_t nwt, bool* ok, hipStream_t stream) {
    *ok = false;
    if (b->tile_recs_n + n > b->tile_recs_cap) {
        size_t free_b = 0, total_b = 0;
        KCHECK_HIP(hipMemGetInfo(&free_b, &total_b));
        uint64_t pool[3] = {0, 0, 0};
        dev_cache_stats(b->s.device, pool);
        free_b += pool[1];                         // (what the caching allocator holds idle is as good as free)
        // room for sixteen batches like this one to begin with (a build is a dozen batches), doubled when that was too little
        const uint64_t want = std::max<uint64_t>(b->tile_recs_cap ? b->tile_recs_cap * 2 : n * 16, b->tile_recs_n + n);
        // (the counting needs the records twice more -- the passes' scratch and the next level)
        // (KATOME_TILE_RECS_LIMIT: no more records than this are kept aside -- tests; by default a sixth of the card's memory)
        static const uint64_t limit = getenv("KATOME_TILE_RECS_LIMIT") ? strtoull(getenv("KATOME_TILE_RECS_LIMIT"), nullptr, 10) : ~0ull;
        if (b->tile_recs_n + n > limit || want * 8 * nwt * 3 > free_b + b->tile_recs_cap * 8 * nwt || want * 8 * nwt > total_b / 6) {
            KCHECK(flush_tile_recs(b, stream));
            return KATOME_OK;
        }
        DevBuf grown(stream);
        if (grown.alloc(want * 8 * nwt + 16) != KATOME_OK) {          // (no room after all: the table takes over)
            KCHECK(flush_tile_recs(b, stream));
            return KATOME_OK;
        }
        if (b->tile_recs_n) KCHECK_HIP(hipMemcpyAsync(grown.p, b->tile_recs.p, b->tile_recs_n * 8 * nwt, hipMemcpyDeviceToDevice, stream));
        const size_t grown_bytes = grown.bytes;
        b->tile_recs.adopt(grown.take(), grown_bytes);
        b->tile_recs.stream = stream;
        b->tile_recs_cap = want;
    }
    if (b->tile_recs_n == 0) b->tile_recs_exact = true;          // (nothing kept yet: the host knows the count until a batch with holes comes)
    *ok = true;
    return KATOME_OK;
}

// a batch's tiles kept aside as records; *kept = false: they go into the tile table
int keep_tile_recs(katome_builder* b, const uint64_t* d_records, uint64_t n, uint32_t nwt, bool* kept, hipStream_t stream) {
    *kept = false;
    bool ok = false;
    // (first-seen order: a record is kept with its tag -- read << 32 | number of its first window << 16 | of its reverse complement's)
    const uint32_t words = nwt + (b->first_seen ? 1 : 0);
    KCHECK(reserve_tile_recs(b, n, words, &ok, stream));
    if (!ok) return KATOME_OK;
    if (!b->tile_recs_count.p) KCHECK(b->tile_recs_count.alloc(8, stream));
    if (b->tile_recs_exact) {
        // from here on only the device knows how many of the records handed over were valid: its cursor starts at what the host knew
        const uint64_t start = b->tile_recs_n;
        KCHECK_HIP(hipMemcpyAsync(b->tile_recs_count.p, &start, 8, hipMemcpyHostToDevice, stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        b->tile_recs_exact = false;
    }
    if (b->first_seen) {
        const uint32_t W = b->seen_read_len - b->s.k + 1;
        KCHECK(table_keep_rest(d_records, n, nwt, true, b->reads_inserted, W / b->span, 0, 2 * W, b->tile_recs.as<u64>(), b->tile_recs_count.as<u64>(), stream,
                               b->span, b->span));
    } else
    KCHECK(table_keep_rest(d_records, n, nwt, false, 0, 1, 0, 0, b->tile_recs.as<u64>(), b->tile_recs_count.as<u64>(), stream));      // (the valid ones, behind the cursor)
    b->tile_recs_n += n;
    *kept = true;
    return KATOME_OK;
}

// Device memory a counting level may plan with: what the driver has free plus what the library's cache holds idle
// (KATOME_LEVEL_BUDGET: no more than this many bytes -- tests of the ways back into the tables at sizes the oracle can check)
static uint64_t level_budget() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return ~0ull; }
    const uint64_t avail = (uint64_t)free_b + dev_cached_bytes();
    const char* e = getenv("KATOME_LEVEL_BUDGET");
    return e ? std::min<uint64_t>(avail, strtoull(e, nullptr, 10)) : avail;
}
// a level counted by sorting holds its records, the partition passes' scratch of the same size and an output list sized for the case
// that no record repeats: three times the records (oriented edges: the output twice over)
static bool level_fits(uint64_t n_records, uint32_t key_words, bool oriented, const char* what) {
    const char* sl = getenv("KATOME_LEVEL_SLACK");           // (room left for everything else: group index, cursors, the allocator's rounding)
    const uint64_t slack = sl ? strtoull(sl, nullptr, 10) : (256ull << 20);
    const uint64_t pair = 8ull * key_words + 4, need = n_records * pair * (oriented ? 4 : 3) + slack, have = level_budget();
    if (need <= have) return true;
    if (getenv("KATOME_LEVEL_TRACE"))
        fprintf(stderr, "[katome levels] %s: %llu records need %.1f GiB by sorting, %.1f GiB available -- this level and those below are counted in tables\n",
                what, (unsigned long long)n_records, need / 1073741824.0, have / 1073741824.0);
    return false;
}
// a list that was sized for its records and holds far fewer keys moves into a buffer of its own size (thin coverage: the room is needed)
static int shrink_to_fit(DevBuf& buf, size_t used, hipStream_t stream) {
    if (!buf.p || buf.bytes < (1ull << 30) || used * 2 > buf.bytes) return KATOME_OK;
    DevBuf small(stream);
    if (small.alloc(used + 64) != KATOME_OK) return KATOME_OK;          // (no room for the copy: the list stays where it is)
    if (used) KCHECK_HIP(hipMemcpyAsync(small.p, buf.p, used, hipMemcpyDeviceToDevice, stream));
    const size_t n = small.bytes;
    buf.adopt(small.take(), n);
    buf.stream = stream;
    return KATOME_OK;
}

// The k-mer level of an input whose tiles hardly repeat, when its records do not fit the card at once (thin coverage at hundreds of
// millions of reads): the last tile level's list is taken in P parts -- a part's tiles -> their k-mer records -> counted by sorting,
// strands not yet told apart, into a list of (canonical k-mer, count) --, and the parts' lists, put behind one another, ARE weighted
// k-mer records again: the caller counts them once more, as it would have counted the level's records, and has its edges.  A k-mer
// that several parts hold is counted 1 + 1/P times over instead of once; the alternative is the k-mer table at a third of the speed
// (profiles/r04_coverage_sweep.jsonl).  KATOME_E_UNSUPPORTED: not this way either (the caller goes on in tables; the list is untouched).
// KATOME_LEVEL_PARTS=0: never
static int kmer_records_in_parts(katome_builder* b, const uint64_t* lk, const uint32_t* lw, uint64_t n_last, uint32_t last_bases, uint32_t last_span,
                                 DevBuf& keys, DevBuf& weights, uint64_t* n_records, uint64_t extra_room, hipStream_t stream) {
    static const bool on = !getenv("KATOME_LEVEL_PARTS") || atoi(getenv("KATOME_LEVEL_PARTS")) != 0;
    if (!on) return KATOME_E_UNSUPPORTED;
    const uint32_t nw = b->nw, k = b->s.k, nwl = (uint32_t)key_words_for_k(last_bases);
    const uint64_t pair = 8ull * nw + 4, total = n_last * last_span, have = level_budget();
    const char* sl = getenv("KATOME_LEVEL_SLACK");
    const uint64_t slack = sl ? strtoull(sl, nullptr, 10) : (256ull << 20);
    // a part's records, the passes' scratch and its list (three times the records) in half of what is there; the other half holds the lists
    uint32_t P = 2;
    while (P <= 64 && (total / P + last_span) * pair * 3 + slack > have / 2) P *= 2;
    if (P > 64 || n_last < P) return KATOME_E_UNSUPPORTED;
    if (getenv("KATOME_LEVEL_TRACE")) fprintf(stderr, "[katome levels] k-mers: counted in %u parts of %llu records\n", P, (unsigned long long)(total / P));
    PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
    std::unique_ptr<DevBuf[]> pk(new DevBuf[P]), pw(new DevBuf[P]);
    for (uint32_t p = 0; p < P; ++p) { pk[p].stream = stream; pw[p].stream = stream; }
    std::vector<uint64_t> pn(P, 0);
    uint64_t n_p = 0;
    for (uint32_t p = 0; p < P; ++p) {
        const uint64_t lo = n_last * p / P, hi = n_last * (p + 1) / P;
        if (hi == lo) continue;
        DevBuf rk(stream), rw(stream);
        uint64_t n_rec = 0, ne = 0, nd = 0;
        int rc = table_list_to_records(lk + lo * nwl, lw + lo, hi - lo, last_bases, k, last_span, 1, b->rc, rk, rw, &n_rec, stream, 0, nullptr);
        if (rc == KATOME_OK) rc = records_to_edges_sorted(rk, rw, n_rec, k, false, 0, pk[p], pw[p], &ne, &nd, stream);
        if (rc == KATOME_E_OOM || rc == KATOME_E_UNSUPPORTED) return KATOME_E_UNSUPPORTED;          // (less room than was planned with: the tables)
        if (rc != KATOME_OK) return rc;
        rk.release(); rw.release();
        KCHECK(shrink_to_fit(pk[p], ne * 8 * nw, stream)); KCHECK(shrink_to_fit(pw[p], ne * 4, stream));
        pn[p] = ne; n_p += ne;
    }
    if (!level_fits(n_p + extra_room, nw, b->rc, "k-mers (the parts' lists)")) return KATOME_E_UNSUPPORTED;
    if (keys.alloc((n_p + extra_room + 1) * 8 * nw, stream) != KATOME_OK || weights.alloc((n_p + extra_room + 1) * 4, stream) != KATOME_OK) {
        keys.release(); weights.release();
        return KATOME_E_UNSUPPORTED;
    }
    uint64_t at = 0;
    for (uint32_t p = 0; p < P; ++p) {
        if (!pn[p]) continue;
        KCHECK_HIP(hipMemcpyAsync(keys.as<u64>() + at * nw, pk[p].p, pn[p] * 8 * nw, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(weights.as<u32>() + at, pw[p].p, pn[p] * 4, hipMemcpyDeviceToDevice, stream));
        at += pn[p];
        pk[p].release(); pw[p].release();
    }
    *n_records = n_p;
    return KATOME_OK;
}

// The tile records kept aside -> the (k-mer, count) records of the last tile level, every level counted by sorting (DESIGN.md
// section 4): records -> two hash passes -> counted in LDS -> a compact list of distinct tiles with their counts; the next level's
// records are cut out of that list.  KATOME_OK: keys / weights hold *n_records records (room for extra_room more behind them) and
// the tile records are gone.  KATOME_E_UNSUPPORTED: a level could not be counted this way (or there is nothing to count) -- the
// tile records, or the distinct big tiles with their counts, are in the tile table instead and the caller goes on in tables.
int tile_recs_to_kmer_records(katome_builder* b, DevBuf& keys, DevBuf& weights, uint64_t* n_records, uint64_t extra_room, hipStream_t stream,
                              DevBuf* first_counts) {
    *n_records = 0;
    const uint32_t k = b->s.k, span = b->span, tile_bases = k + span - 1, nwt = (uint32_t)key_words_for_k(tile_bases);
    b->span2 = mid_span(span);
    DevBuf t1k(stream), t1w(stream);
    uint64_t n1 = 0, d1 = 0;
    int rc;
    {
        PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
        TileLevelScope tl;
        DevBuf ones(stream);          // (stays empty: records without weights count once each, and the passes move 16 bytes a record, not 20)
        uint64_t n = 0;
        KCHECK(tile_recs_valid(b, &n, stream));
        rc = n ? records_to_edges_sorted(b->tile_recs, ones, n, tile_bases, false, 0, t1k, t1w, &n1, &d1, stream)
               : KATOME_E_UNSUPPORTED;                          // (every read was skipped: nothing to count)
        if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
    }
    if (rc == KATOME_E_UNSUPPORTED) {
        KCHECK(flush_tile_recs(b, stream));          // (the records are all still there, in another order: into the table with them)
        return KATOME_E_UNSUPPORTED;
    }
    b->tile_recs.release(); b->tile_recs_count.release(); b->tile_recs_n = b->tile_recs_cap = 0;
    b->tile_recs_exact = false;
    b->tile_recs_closed = true;
    b->stat_tiles = n1; b->stat_tile_slots = 0; b->stat_tiles2 = 0; b->stat_tile2_slots = 0;
    KCHECK(shrink_to_fit(t1k, n1 * 8 * nwt, stream)); KCHECK(shrink_to_fit(t1w, n1 * 4, stream));
    const uint64_t* lk = t1k.as<u64>(); const uint32_t* lw = t1w.as<u32>();
    uint64_t n_last = n1; uint32_t last_bases = tile_bases, last_span = span;
    DevBuf t2k(stream), t2w(stream);
    if (b->span2 && n1) {
        const uint32_t kk2 = k + b->span2 - 1, n_sub = span / b->span2;
        PhaseScope ps(b->prof, PH_EXPAND_MID, stream);
        TileLevelScope tl;
        DevBuf mk(stream), mw(stream);
        uint64_t n_mid = 0, n2 = 0, d2 = 0;
        DevBuf mid_counts(stream);        // (the first partition pass's digit counts per tile, made while the records are written)
        // (input whose tiles hardly repeat -- thin coverage -- multiplies records level by level: a level that would not fit the card
        // by sorting is counted in a table, which takes its upserts in slot ranges of any size)
        if (sorted_fail("mid") || !level_fits(n1 * n_sub, (uint32_t)key_words_for_k(kk2), false, "mid tiles")) rc = KATOME_E_UNSUPPORTED;
        else {
            KCHECK(table_list_to_records(lk, lw, n1, tile_bases, kk2, n_sub, b->span2, b->rc, mk, mw, &n_mid, stream, 0, &mid_counts));
            rc = records_to_edges_sorted(mk, mw, n_mid, kk2, false, 0, t2k, t2w, &n2, &d2, stream, nullptr, mid_counts.as<u32>());
        }
        if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
        if (rc == KATOME_E_UNSUPPORTED) {
            // the mid tiles cannot be counted this way: the distinct big tiles go into the tile table with their counts, and the build
            // goes on from there as if they had been counted in it
            mk.release(); mw.release(); t2k.release(); t2w.release();
            KCHECK(builder_insert(b, b->tiles, b->tiles_ready, nwt, b->s.table_slots_hint / 4, t1k.as<u64>(), t1w.as<u32>(), n1, nullptr, PH_INSERT_TILES, stream));
            return KATOME_E_UNSUPPORTED;
        }
        b->stat_tiles2 = n2;
        t1k.release(); t1w.release();
        mk.release(); mw.release();
        KCHECK(shrink_to_fit(t2k, n2 * 8 * key_words_for_k(kk2), stream)); KCHECK(shrink_to_fit(t2w, n2 * 4, stream));
        lk = t2k.as<u64>(); lw = t2w.as<u32>(); n_last = n2; last_bases = kk2; last_span = b->span2;
    }
    if (n_last && !level_fits(n_last * last_span + extra_room, b->nw, b->rc, "k-mers")) {
        // in parts first (kmer_records_in_parts): their lists stand in for the level's records
        if (first_counts) first_counts->release();
        const int prc = kmer_records_in_parts(b, lk, lw, n_last, last_bases, last_span, keys, weights, n_records, extra_room, stream);
        if (prc != KATOME_E_UNSUPPORTED) return prc;
        keys.release(); weights.release(); *n_records = 0;
        // the k-mer level would not fit by sorting: the distinct tiles of the last level go into their table with their counts (the mid
        // tiles into `tiles2`, with nothing in `tiles`: expand_to_last_level's "done before" state) and the k-mers are counted in theirs
        const uint32_t nwl = (uint32_t)key_words_for_k(last_bases);
        if (b->span2) {
            KCHECK(builder_insert(b, b->tiles2, b->tiles2_ready, nwl, std::max<uint64_t>(b->s.table_slots_hint / 4, n_last * 2), lk, lw, n_last, nullptr, PH_EXPAND_MID, stream));
            b->tiles_ready = true;                   // (with b->tiles empty)
            b->stat_tile2_slots = b->tiles2.cap;
        } else {
            KCHECK(builder_insert(b, b->tiles, b->tiles_ready, nwl, b->s.table_slots_hint / 4, lk, lw, n_last, nullptr, PH_INSERT_TILES, stream));
        }
        return KATOME_E_UNSUPPORTED;
    }
    PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
    // (first_counts: for a caller that orders exactly these records by the whole k-mer's hash next -- table_list_to_records)
    return table_list_to_records(lk, lw, n_last, last_bases, k, last_span, 1, b->rc, keys, weights, n_records, stream, extra_room, first_counts);
}

// keeps a batch's left-over windows aside (see builder.h); *kept = false: they have to go into the table
static int keep_rest(katome_builder* b, const uint64_t* d_records, uint64_t n, bool* kept, hipStream_t stream) {
    *kept = false;
    const uint32_t words = b->nw + (b->first_seen ? 1 : 0);          // (first-seen order: tagged records)
    if (b->rest_n + n > b->rest_cap) {
        size_t free_b = 0, total_b = 0;
        KCHECK_HIP(hipMemGetInfo(&free_b, &total_b));
        const uint64_t want = std::max<uint64_t>(b->rest_n + n, b->rest_cap * 2);
        // (they are sorted with the tiles' k-mers later: past an eighth of the card they stop being a side matter)
        if ((want + b->rest_n) * 8 * words > free_b / 2 || want * 8 * words > total_b / 8) { KCHECK(flush_rest(b, stream)); return KATOME_OK; }
        DevBuf grown(stream);
        KCHECK(grown.alloc(want * 8 * words + 16));
        if (b->rest_n) KCHECK_HIP(hipMemcpyAsync(grown.p, b->rest_k.p, b->rest_n * 8 * words, hipMemcpyDeviceToDevice, stream));
        const size_t grown_bytes = grown.bytes;
        b->rest_k.adopt(grown.take(), grown_bytes);
        b->rest_k.stream = stream;
        b->rest_cap = want;
    }
    if (!b->rest_count.p) { KCHECK(b->rest_count.alloc(8, stream)); KCHECK_HIP(hipMemsetAsync(b->rest_count.p, 0, 8, stream)); }
    KCHECK(table_keep_rest(d_records, n, b->nw, b->first_seen, b->last_batch_read0, b->rem_per_read, b->rem_win0,
                           b->first_seen ? 2u * (b->seen_read_len - b->s.k + 1) : 0u, b->rest_k.as<u64>(), b->rest_count.as<u64>(), stream));
    b->rest_n += n;
    *kept = true;
    return KATOME_OK;
}

extern "C" {

int katome_dev_insert_weighted(katome_builder* b, const uint64_t* d_records, const uint32_t* d_weights, uint64_t n_records,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (b->edges_ready) { set_error("builder already finalized"); return KATOME_E_ARG; }
    if (n_records == 0) {
        if (b->first_seen && b->var_prefix && b->var_mode != 1 && b->var_records == 0) {   // a batch whose reads are all whole tiles
            b->var_seq_base += 2 * b->var_windows;
            b->var_prefix = b->var_rec_prefix = nullptr; b->var_reads = b->var_windows = 0;
        }
        return KATOME_OK;
    }
    if (!b->first_seen && !d_weights && (b->tiles_ready || b->tile_recs_n) && !b->table_ready && !b->rest_closed && b->nw <= 2 && sorted_count_mode()) {
        bool kept = false;
        KCHECK(keep_rest(b, d_records, n_records, &kept, stream));
        if (kept) return KATOME_OK;
    }
    // first-seen order, reads of one length: the windows after the batch's tiles wait as tagged records (table.hip, seen_pack)
    if (b->first_seen && b->rem_pending && !d_weights && !b->var_prefix && !b->var_seq_base && (b->tiles_ready || b->tile_recs_n) && !b->table_ready && !b->rest_closed &&
        b->nw <= 2 && sorted_count_mode() && b->seen_read_len >= b->s.k && 2ull * (b->seen_read_len - b->s.k + 1) <= 0xFFFFu &&
        b->last_batch_read0 + b->last_batch_reads < (1ull << 32) && n_records == b->last_batch_reads * b->rem_per_read) {
        bool kept = false;
        KCHECK(keep_rest(b, d_records, n_records, &kept, stream));
        if (kept) { b->rem_pending = false; return KATOME_OK; }
    }
    uint64_t room = 0;
    KCHECK(ensure_table(b, n_records, &room, stream));
    // Optional: order the batch by table region first (streaming radix passes over the hash prefix), so
    // that the insert kernel's working set stays cache-sized.  Off unless KATOME_REGION_PASSES says so:
    // measured on MI355X the insert kernel is bound by atomic throughput, not by where the slots live.
    const uint64_t* k_in = d_records; const uint32_t* w_in = d_weights;
    SeenOrigin origin;
    if (b->first_seen && b->var_prefix) {
        if (b->var_mode == 1 || n_records != b->var_records) { set_error("first-seen order: insert exactly the records the last katome_dev_extract_var* call produced"); return KATOME_E_ARG; }
        origin.win_prefix = b->var_prefix; origin.n_reads = b->var_reads; origin.seq_base = b->var_seq_base; origin.rc = b->rc;
        origin.rec_prefix = b->var_rec_prefix; origin.mode = b->var_mode; origin.span = b->var_span;
    } else if (b->first_seen) {
        if (b->var_seq_base) { set_error("first-seen order: fixed- and variable-length batches cannot be mixed in one build"); return KATOME_E_UNSUPPORTED; }
        if (b->seen_read_len < b->s.k) { set_error("first-seen order: extract the records with this builder first"); return KATOME_E_ARG; }
        origin.windows = b->seen_read_len - b->s.k + 1; origin.span = 1; origin.rc = b->rc;
        if (b->rem_pending) {           // the windows after the tiles of the batch that was just inserted
            origin.per_read = b->rem_per_read; origin.win0 = b->rem_win0; origin.read0 = b->last_batch_read0;
            if (n_records != b->last_batch_reads * origin.per_read) { set_error("first-seen order: insert the remainder of the batch whose tiles were inserted last"); return KATOME_E_ARG; }
        } else {
            origin.per_read = origin.windows; origin.read0 = b->reads_inserted;
            if (n_records % origin.per_read) { set_error("first-seen order: a batch must hold whole reads"); return KATOME_E_ARG; }
        }
    }
    const int passes = b->first_seen ? 0 : region_passes(b->table.cap * b->table.slot_bytes());
    if (passes > 0 && n_records >= (1u << 16)) {
        for (int i = 0; i < 2; ++i) {
            if ((i == 0 || passes > 1) && b->scratch_k[i].bytes < n_records * 8 * b->nw) KCHECK(b->scratch_k[i].alloc(n_records * 8 * b->nw, stream));
            if (d_weights && (i == 0 || passes > 1) && b->scratch_w[i].bytes < n_records * 4) KCHECK(b->scratch_w[i].alloc(n_records * 4, stream));
        }
        PhaseScope ps(b->prof, PH_REGION_ORDER, stream);
        KCHECK(dev_region_order(d_records, d_weights, n_records, b->nw, passes, b->scratch_k[0].as<u64>(), b->scratch_k[1].as<u64>(),
                                b->scratch_w[0].as<u32>(), b->scratch_w[1].as<u32>(), &k_in, &w_in, stream));
    }
    for (uint64_t done = 0; done < n_records;) {
        if (done) KCHECK(ensure_table(b, n_records - done, &room, stream));
        const uint64_t n = std::min(n_records - done, room);
        PhaseScope ps(b->prof, PH_INSERT, stream);
        origin.rec0 = done;
        KCHECK(table_insert(b->table, k_in + done * b->nw, w_in ? w_in + done : nullptr, n, stream, b->first_seen ? &origin : nullptr));
        done += n;
    }
    if (b->first_seen && origin.win_prefix) {
        b->var_seq_base += 2 * b->var_windows;          // the batch is complete: its reads' windows, both strands
        b->var_prefix = b->var_rec_prefix = nullptr; b->var_reads = b->var_windows = 0;
    } else if (b->first_seen && !b->rem_pending) {
        b->reads_inserted += n_records / origin.per_read;
    }
    b->rem_pending = false;
    return KATOME_OK;
}
int katome_dev_insert(katome_builder* b, const uint64_t* d_records, uint64_t n_records, void* stream) {
    return katome_dev_insert_weighted(b, d_records, nullptr, n_records, stream);
}

// may this builder keep a batch's tile records aside and count them by sorting at the end?
static bool keeps_tile_recs(const katome_builder* b, uint32_t nwt) {
    // (the reference's numbering: reads of one length whose numbers pack into a record word -- lds_count_seen_kernel's rule)
    const bool seen_ok = !b->first_seen || (!b->var_prefix && !b->var_seq_base && !b->direct_edges && b->seen_read_len >= b->s.k &&
                                            2ull * (b->seen_read_len - b->s.k + 1) <= 0xFFFFull && (b->reads_inserted >> 31) == 0);
    return seen_ok && tile_recs_shape(nwt, b->nw, b->first_seen) && !b->tiles_ready && !b->table_ready && !b->tile_recs_closed && sorted_count_mode() &&
           sorted_tiles_mode() == 2;
}

int katome_dev_count_tiles(katome_builder* b, const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len, uint32_t span,
                           const uint8_t* d_skip, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!b || (!d_packed && n_reads)) { set_error("null argument"); return KATOME_E_ARG; }
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (b->edges_ready) { set_error("builder already finalized"); return KATOME_E_ARG; }
    if (read_len < b->s.k || span < 2 || b->s.k + span - 1 > 95) { set_error("bad tile span %u", span); return KATOME_E_ARG; }
    const uint32_t per_read = (read_len - b->s.k + 1) / span, nwt = (uint32_t)key_words_for_k(b->s.k + span - 1);
    const uint64_t n = n_reads * per_read;
    if (n == 0) return KATOME_OK;
    // No read skipped and every record kept so far valid: the records are made where they are kept -- no buffer in between, no copy
    // (C3: 12.8 GB not read and not written again per build).  Otherwise: into a scratch of the builder's, then as
    // katome_dev_insert_tiles does.
    if (!d_skip && !b->first_seen && keeps_tile_recs(b, nwt) && (b->tile_recs_n == 0 || (b->tile_recs_exact && span == b->span))) {
        bool ok = false;
        {
            PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
            KCHECK(reserve_tile_recs(b, n, nwt, &ok, stream));
        }
        if (ok && b->tile_recs_exact) {
            KCHECK(katome_dev_extract_tiles(b, d_packed, n_reads, read_len, span, nullptr, b->tile_recs.as<u64>() + b->tile_recs_n * nwt, stream));
            b->span = span;
            b->tile_recs_n += n;
            return KATOME_OK;
        }
    }
    KCHECK(b->tile_scratch.alloc(n * 8 * nwt + 64, stream));
    KCHECK(katome_dev_extract_tiles(b, d_packed, n_reads, read_len, span, d_skip, b->tile_scratch.as<u64>(), stream));
    return katome_dev_insert_tiles(b, b->tile_scratch.as<u64>(), n, span, stream);
}

int katome_dev_insert_tiles(katome_builder* b, const uint64_t* d_records, uint64_t n_records, uint32_t span, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (b->edges_ready) { set_error("builder already finalized"); return KATOME_E_ARG; }
    if (span < 2 || b->s.k + span - 1 > 95 || ((b->tiles_ready || b->tile_recs_n) && span != b->span)) { set_error("bad tile span %u", span); return KATOME_E_ARG; }
    if (b->first_seen && b->var_prefix && b->s.k + span - 1 > 95) { set_error("variable-length reads: tiles of at most 95 bases"); return KATOME_E_ARG; }
    if (n_records == 0) return KATOME_OK;
    b->span = span;
    const uint32_t nwt = (uint32_t)key_words_for_k(b->s.k + span - 1);
    if (keeps_tile_recs(b, nwt)) {
        const uint32_t per_read = b->first_seen ? (b->seen_read_len - b->s.k + 1) / span : 1;
        if (b->first_seen && (per_read == 0 || n_records % per_read)) { set_error("first-seen order: a batch must hold whole reads"); return KATOME_E_ARG; }
        bool kept = false;
        PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
        KCHECK(keep_tile_recs(b, d_records, n_records, nwt, &kept, stream));
        if (kept) {
            if (b->first_seen) {
                b->last_batch_read0 = b->reads_inserted; b->last_batch_reads = n_records / per_read;
                b->reads_inserted += n_records / per_read;
            }
            return KATOME_OK;
        }
    }
    SeenOrigin origin;
    const bool var_tiles = b->first_seen && b->var_prefix != nullptr;
    if (var_tiles) {
        if (b->var_mode != 1 || b->var_span != span || n_records != b->var_records) { set_error("first-seen order: insert exactly the tiles katome_dev_extract_var_tiles produced"); return KATOME_E_ARG; }
        origin.win_prefix = b->var_prefix; origin.n_reads = b->var_reads; origin.seq_base = b->var_seq_base; origin.rc = b->rc;
        origin.rec_prefix = b->var_rec_prefix; origin.mode = 1; origin.span = span;
    } else if (b->first_seen) {
        if (b->var_seq_base) { set_error("first-seen order: fixed- and variable-length batches cannot be mixed in one build"); return KATOME_E_UNSUPPORTED; }
        if (b->seen_read_len < b->s.k) { set_error("first-seen order: extract the records with this builder first"); return KATOME_E_ARG; }
        origin.windows = b->seen_read_len - b->s.k + 1; origin.per_read = origin.windows / span; origin.span = span; origin.rc = b->rc;
        origin.read0 = b->reads_inserted;
        if (origin.per_read == 0 || n_records % origin.per_read) { set_error("first-seen order: a batch must hold whole reads"); return KATOME_E_ARG; }
    }
    for (uint64_t done = 0; done < n_records;) {
        uint64_t room = 0;
        KCHECK(ensure_table(b, b->tiles, b->tiles_ready, nwt, b->s.table_slots_hint / 4, n_records - done, &room, stream));
        const uint64_t n = std::min(n_records - done, room);
        PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
        origin.rec0 = done;
        KCHECK(table_insert(b->tiles, d_records + done * nwt, nullptr, n, stream, b->first_seen ? &origin : nullptr));
        done += n;
    }
    if (b->first_seen && !var_tiles) {
        b->last_batch_read0 = b->reads_inserted; b->last_batch_reads = n_records / origin.per_read;
        b->reads_inserted += n_records / origin.per_read;
    }
    return KATOME_OK;
}

/* multi-GPU: the (k-mer, weight) records of this rank's tiles, to be routed to the k-mers' owners */
int katome_dev_expand_tiles(katome_builder* b, uint64_t** d_keys, uint32_t** d_weights, uint64_t* n_records, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    *n_records = 0; *d_keys = nullptr; *d_weights = nullptr;
    if (b->first_seen) { set_error("first-seen order is not available on the multi-GPU route"); return KATOME_E_UNSUPPORTED; }
    if (b->tile_recs_n) KCHECK(flush_tile_recs(b, stream));
    if (b->rest_n) KCHECK(flush_rest(b, stream));
    if (!b->tiles_ready) return KATOME_OK;
    Table* last = nullptr; uint32_t last_span = 1;
    KCHECK(expand_to_last_level(b, &last, &last_span, stream));
    {
        PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
        KCHECK(table_expand_tiles_to_records(*last, b->s.k, last_span, b->rc, b->scratch_k[0], b->scratch_w[0], n_records, stream));
    }
    b->tiles.release();
    b->tiles2.release();
    b->tiles_ready = false; b->tiles2_ready = false;
    *d_keys = b->scratch_k[0].as<u64>(); *d_weights = b->scratch_w[0].as<u32>();
    return KATOME_OK;
}

static int weak_edges_ordered(katome_builder* b, uint32_t threshold, hipStream_t stream) {
    PruneGraph g{&b->edge_src, &b->edge_dst, &b->edge_weight, &b->edge_key, &b->node_key, &b->edge_age, b->n_edges, b->n_nodes, b->nw};
    KCHECK(dev_remove_weak_edges_ordered(g, threshold, stream));
    b->n_edges = g.n_edges; b->n_nodes = g.n_nodes;
    return KATOME_OK;
}

int katome_dev_remove_weak_edges(katome_builder* b, uint32_t threshold, void* stream_) {
    if (!b) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    if (b->finalized && b->first_seen) {
        // on the finished graph, with petgraph's numbering (retain_edges / retain_nodes: descending swap_removes)
        KCHECK_HIP(hipSetDevice(b->s.device));
        KCHECK(weak_edges_ordered(b, threshold, stream));
        KCHECK(dev_labels(b->edge_key.as<u64>(), b->n_edges, b->s.k, b->edge_label.as<uint8_t>(), stream));
        KCHECK_HIP(hipStreamSynchronize(stream));
        return KATOME_OK;
    }
    if (b->edges_ready) { set_error("builder already finalized"); return KATOME_E_ARG; }
    b->prune_weight = threshold;
    return KATOME_OK;
}

int katome_dev_table_count(katome_builder* b, uint64_t* out, void* stream) {
    KCHECK_HIP(hipSetDevice(b->s.device));
    *out = 0;
    if (b->rest_n) KCHECK(flush_rest(b, (hipStream_t)stream));
    if (!b->table_ready) return KATOME_OK;
    return table_occupied(b->table, out, (hipStream_t)stream);
}

int katome_dev_edges(katome_builder* b, uint64_t** d_edge_key, uint32_t** d_edge_weight, uint64_t* n_edges, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (!b->edges_ready) {
        b->n_edges = 0;
        b->tile_scratch.release();
        // Counting the last level by sorting instead of in a table (table.hip, lds_count_kernel / lds_count_wide_kernel): k <= 63, by
        // packed key, nothing in the k-mer table yet (left-over windows were kept aside), enough tiles to be worth the extra launches
        const int sorted_count = sorted_count_mode();
        bool counted = false;
        // ... and the tile levels above it the same way when the tiles were kept as records (katome_dev_insert_tiles): every level is
        // "records -> two hash passes -> counted in LDS -> a compact list of distinct keys with their counts", the next level's
        // records are cut out of that list (table.hip, list_to_records_kernel)
        if (!b->first_seen && b->tile_recs_n && (b->table_ready || (b->tile_recs_n * b->span < (1ull << 22) && sorted_count != 2)))
            KCHECK(flush_tile_recs(b, stream));      // (k-mers in the table already, or too few tiles to be worth it)
        if (!b->first_seen && b->tile_recs_n) {
            const uint32_t k = b->s.k;
            DevBuf rk(stream), rw(stream);
            uint64_t n_rec = 0, n_rest = 0, distinct = 0;
            KCHECK(rest_valid(b, &n_rest, stream));
            DevBuf first_counts(stream);
            int rc = tile_recs_to_kmer_records(b, rk, rw, &n_rec, n_rest, stream, &first_counts);
            if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
            if (rc == KATOME_OK) {         // (otherwise the tiles are in their table now, and the blocks below take it from there)
                {
                    PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
                    if (n_rest) {
                        KCHECK_HIP(hipMemcpyAsync(rk.as<u64>() + n_rec * b->nw, b->rest_k.p, n_rest * 8 * b->nw, hipMemcpyDeviceToDevice, stream));
                        KCHECK(dev_fill_u32(rw.as<u32>() + n_rec, n_rest, 1u, stream));
                        n_rec += n_rest;
                    }
                    rest_reset(b);
                    rc = sorted_fail("last") ? KATOME_E_UNSUPPORTED
                        : records_to_edges_sorted(rk, rw, n_rec, k, b->rc, b->prune_weight, b->edge_key, b->edge_weight, &b->n_edges, &distinct, stream, nullptr,
                                                  n_rest ? nullptr : first_counts.as<u32>());
                    if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
                }
                if (rc == KATOME_OK) {
                    b->stat_kmers = distinct; b->stat_kmer_slots = 0;
                    rk.release(); rw.release();
                    PhaseScope ps(b->prof, PH_SORT_EDGES, stream);
                    KCHECK(dev_sort_bufs(b->edge_key, &b->edge_weight, b->n_edges, b->nw, 2 * k, stream, true));
                    counted = true;
                } else {
                    // the k-mers cannot be counted this way (a hash group beyond the LDS route): their records, counts and all, go
                    // into the k-mer table and the edges are read out of it
                    b->n_edges = 0;
                    KCHECK(builder_insert(b, b->table, b->table_ready, b->nw, b->s.table_slots_hint, rk.as<u64>(), rw.as<u32>(), n_rec, nullptr, PH_INSERT, stream));
                }
            }
        }
        if (!counted && sorted_count && b->tiles_ready && b->tiles.cap && !b->table_ready && !b->first_seen && b->nw <= 2) {
            uint64_t n_tiles = 0;
            KCHECK(table_occupied(b->tiles, &n_tiles, stream));
            const uint64_t bound = n_tiles * b->span + b->rest_n;        // the k-mer records can be no more than this
            const uint32_t nwt = b->tiles.nw;
            const bool shapes = (b->nw == 1 && nwt <= 2) || (b->nw == 2 && (nwt == 2 || nwt == 3));      // (what tiles_to_records streams)
            if (shapes && (bound >= (1ull << 22) || (sorted_count == 2 && bound)) && (bound >> 21) <= 2900) {      // (2: however few -- tests)
                Table* last = nullptr; uint32_t last_span = 1;
                // Big tiles that are in their table (KATOME_SORTED_TILES=1, a build that could not keep them aside as records, or one
                // that had to give them up half-way): the mid tiles are counted by sorting all the same -- the big tiles' sub-tiles
                // leave the tile table as records, two hash passes, counted in LDS into a compact list of (mid tile, count), and the
                // k-mer records are cut out of that list -- no mid-tile table, no 7e8 128-bit upserts (C3: 45 -> 32 ms for the level).
                DevBuf t2k(stream), t2w(stream);
                uint64_t n_mid_list = 0;
                bool mid_sorted = false;
                const uint32_t sp2 = mid_span(b->span);
                if (sorted_tiles_mode() && nwt == 2 && b->nw == 1 && sp2 && !b->tiles2_ready && b->tiles.cap && n_tiles &&
                    level_fits(n_tiles * (b->span / sp2), (uint32_t)key_words_for_k(b->s.k + sp2 - 1), false, "mid tiles (big tiles in their table)")) {
                    const uint32_t kk2 = b->s.k + sp2 - 1, n_sub = b->span / sp2;
                    PhaseScope ps(b->prof, PH_EXPAND_MID, stream);
                    TileLevelScope tl;
                    DevBuf mk(stream), mw(stream);
                    uint64_t n_mid = 0, d2 = 0;
                    KCHECK(table_expand_tiles_to_subtiles(b->tiles, kk2, n_sub, sp2, b->rc, mk, mw, &n_mid, stream, nullptr));
                    int rc2 = (n_mid && !sorted_fail("mid")) ? records_to_edges_sorted(mk, mw, n_mid, kk2, false, 0, t2k, t2w, &n_mid_list, &d2, stream) : KATOME_E_UNSUPPORTED;
                    if (rc2 != KATOME_OK && rc2 != KATOME_E_UNSUPPORTED) return rc2;
                    if (rc2 == KATOME_OK) {
                        mid_sorted = true;           // (the big-tile table stays until the k-mers are counted: the table route's way back)
                        b->stat_tiles = n_tiles; b->stat_tile_slots = b->tiles.cap;
                        b->span2 = sp2; b->stat_tiles2 = n_mid_list; b->stat_tile2_slots = 0;
                        last_span = sp2;
                    } else { t2k.release(); t2w.release(); }
                }
                if (!mid_sorted) KCHECK(expand_to_last_level(b, &last, &last_span, stream));
                bool last_ok = mid_sorted || (b->nw == 1 && last->nw <= 2) || (b->nw == 2 && (last->nw == 2 || last->nw == 3));
                if (last_ok) {        // (... and the records of the level fit the card with their scratch and their output)
                    uint64_t last_tiles = n_mid_list;
                    if (!mid_sorted) KCHECK(table_occupied(*last, &last_tiles, stream));
                    last_ok = level_fits(last_tiles * last_span + b->rest_n, b->nw, b->rc, "k-mers (tiles in their table)");
                }
                if (last_ok) {
                    DevBuf rk(stream), rw(stream);
                    uint64_t n_rec = 0, distinct = 0;
                    int rc = KATOME_OK;
                    {
                        PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
                        uint64_t n_rest = 0;
                        KCHECK(rest_valid(b, &n_rest, stream));
                        if (mid_sorted) {
                            KCHECK(table_list_to_records(t2k.as<u64>(), t2w.as<u32>(), n_mid_list, b->s.k + sp2 - 1, b->s.k, sp2, 1, b->rc, rk, rw, &n_rec, stream, n_rest));
                            t2k.release(); t2w.release();
                        } else
                        KCHECK(table_tiles_to_records_fast(*last, b->s.k, last_span, b->rc, rk, rw, &n_rec, stream, n_rest));
                        if (n_rest) {             // the left-over windows behind them, one each
                            KCHECK_HIP(hipMemcpyAsync(rk.as<u64>() + n_rec * b->nw, b->rest_k.p, n_rest * 8 * b->nw, hipMemcpyDeviceToDevice, stream));
                            KCHECK(dev_fill_u32(rw.as<u32>() + n_rec, n_rest, 1u, stream));
                            n_rec += n_rest;
                        }
                        rc = sorted_fail("last") ? KATOME_E_UNSUPPORTED
                            : records_to_edges_sorted(rk, rw, n_rec, b->s.k, b->rc, b->prune_weight, b->edge_key, b->edge_weight, &b->n_edges, &distinct, stream);
                        if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
                    }
                    if (rc == KATOME_OK) {
                        b->tiles.release(); b->tiles2.release();
                        b->tiles_ready = false; b->tiles2_ready = false;
                        rest_reset(b);
                        b->stat_kmers = distinct; b->stat_kmer_slots = 0;
                        for (int i = 0; i < 2; ++i) { b->scratch_k[i].release(); b->scratch_w[i].release(); }
                        rk.release(); rw.release();
                        PhaseScope ps(b->prof, PH_SORT_EDGES, stream);
                        KCHECK(dev_sort_bufs(b->edge_key, &b->edge_weight, b->n_edges, b->nw, 2 * b->s.k, stream, true));
                        counted = true;
                    } else {
                        b->n_edges = 0;           // (a group too large for the LDS route: the table counts, from the tiles that are still there)
                    }
                }
            }
        }
        // The same for a first-seen-order build (table.hip, lds_count_seen_kernel): reads of one length whose windows are whole tiles
        // (nothing in the k-mer table), so that a record's two sequence numbers pack into one word
        DevBuf raw_w(stream), raw_seq(stream);
        bool counted_seen = false;
        // ... and with the tiles kept as TAGGED records (keep_tile_recs in such a build) every level of it: a level's distinct keys leave
        // with their counts and their two lowered numbers packed into a tag again (lds_count_seen_kernel, LIST), the next level's tagged
        // records are cut out of that list (list_to_tagged_records_kernel: a sub-window's numbers are its tile's plus its place in it)
        if (b->first_seen && b->tile_recs_n && (b->table_ready || (b->tile_recs_n * b->span < (1ull << 22) && sorted_count != 2)))
            KCHECK(flush_tile_recs(b, stream));
        if (b->first_seen && b->tile_recs_n) {
            const uint32_t k = b->s.k, span = b->span, tile_bases = k + span - 1;
            const uint64_t spr = 2ull * (b->seen_read_len - k + 1);
            b->span2 = mid_span(span);
            uint64_t n = 0, n1 = 0, d1 = 0, distinct = 0;
            KCHECK(tile_recs_valid(b, &n, stream));
            DevBuf l1(stream), c1(stream);
            int rc = KATOME_E_UNSUPPORTED;
            if (n) {
                PhaseScope ps(b->prof, PH_INSERT_TILES, stream);
                TileLevelScope tl;
                DevBuf ones(stream);               // (stays empty: the records count once each)
                rc = tagged_records_sorted(b->tile_recs, ones, n, tile_bases, b->rc, spr, true, l1, c1, &n1, &d1, stream);
                if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
            }
            if (rc == KATOME_E_UNSUPPORTED) {
                KCHECK(flush_tile_recs(b, stream));          // (the records are all still there, in another order: into the table with them)
            } else {
                b->tile_recs.release(); b->tile_recs_count.release(); b->tile_recs_n = b->tile_recs_cap = 0;
                b->tile_recs_exact = false; b->tile_recs_closed = true;
                b->stat_tiles = n1; b->stat_tile_slots = 0; b->stat_tiles2 = 0; b->stat_tile2_slots = 0;
                const uint64_t* lk = l1.as<u64>(); const uint32_t* lw = c1.as<u32>();
                uint64_t n_last = n1; uint32_t last_bases = tile_bases, last_span = span;
                DevBuf l2(stream), c2(stream);
                if (b->span2 && n1) {
                    const uint32_t kk2 = k + b->span2 - 1, n_sub = span / b->span2;
                    PhaseScope ps(b->prof, PH_EXPAND_MID, stream);
                    TileLevelScope tl;
                    DevBuf mk(stream), mw(stream);
                    uint64_t n_mid = 0, n2 = 0, d2 = 0;
                    DevBuf mid_counts(stream);         // (the first partition pass's digit counts per tile, made while the records are written)
                    KCHECK(table_list_to_tagged_records(lk, lw, n1, tile_bases, kk2, n_sub, b->span2, b->rc, mk, mw, &n_mid, stream, 0, &mid_counts));
                    rc = sorted_fail("mid") ? KATOME_E_UNSUPPORTED
                        : tagged_records_sorted(mk, mw, n_mid, kk2, b->rc, spr, true, l2, c2, &n2, &d2, stream, mid_counts.as<u32>());
                    if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
                    if (rc == KATOME_OK) { b->stat_tiles2 = n2; lk = l2.as<u64>(); lw = c2.as<u32>(); n_last = n2; last_bases = kk2; last_span = b->span2; }
                }
                if (rc == KATOME_OK) {
                    PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
                    DevBuf kr(stream), kw(stream);
                    uint64_t n_rec = 0, n_rest = 0;
                    KCHECK(rest_valid(b, &n_rest, stream));
                    DevBuf last_counts(stream);
                    KCHECK(table_list_to_tagged_records(lk, lw, n_last, last_bases, k, last_span, 1, b->rc, kr, kw, &n_rec, stream, n_rest, &last_counts));
                    l2.release(); c2.release();
                    if (n_rest) {             // the left-over windows behind them (tagged records already), one each
                        KCHECK_HIP(hipMemcpyAsync(kr.as<u64>() + n_rec * (b->nw + 1), b->rest_k.p, n_rest * 8 * (b->nw + 1), hipMemcpyDeviceToDevice, stream));
                        KCHECK(dev_fill_u32(kw.as<u32>() + n_rec, n_rest, 1u, stream));
                        n_rec += n_rest;
                    }
                    rc = sorted_fail("last") ? KATOME_E_UNSUPPORTED
                        : tagged_records_sorted(kr, kw, n_rec, k, b->rc, spr, false, b->edge_key, raw_seq, &b->n_edges, &distinct, stream, last_counts.as<u32>());
                    if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
                }
                if (rc == KATOME_OK) {
                    l1.release(); c1.release();
                    rest_reset(b);
                    b->stat_kmers = distinct; b->stat_kmer_slots = 0;
                    counted_seen = true;
                } else {
                    // a level below gave up: the distinct big tiles go into the tile table with their counts and their numbers, and the
                    // build goes on from there as if they had been counted in it
                    b->n_edges = 0;
                    l2.release(); c2.release();
                    const uint32_t nwt = (uint32_t)key_words_for_k(tile_bases);
                    DevBuf keys(stream), pairs(stream);
                    KCHECK(keys.alloc(n1 * 8 * nwt + 16)); KCHECK(pairs.alloc(n1 * 16 + 16));
                    KCHECK(table_tagged_to_pairs(l1.as<u64>(), n1, nwt, spr, keys.as<u64>(), pairs.as<u64>(), stream));
                    l1.release();
                    SeenOrigin origin;
                    origin.pairs = pairs.as<u64>(); origin.rc = b->rc;
                    KCHECK(builder_insert(b, b->tiles, b->tiles_ready, nwt, b->s.table_slots_hint / 4, keys.as<u64>(), c1.as<u32>(), n1, &origin, PH_INSERT_TILES, stream));
                }
            }
        }
        if (!counted_seen && sorted_count && b->first_seen && b->tiles_ready && !b->table_ready && b->nw <= 2 && !b->var_seq_base && !b->var_prefix && !b->direct_edges &&
            b->seen_read_len >= b->s.k) {
            uint64_t n_tiles = 0;
            KCHECK(table_occupied(b->tiles, &n_tiles, stream));
            const uint64_t bound = n_tiles * b->span;
            if (bound >= (1ull << 22) || (sorted_count == 2 && bound)) {       // (the last level's own size is checked where its records are made)
                Table* last = nullptr; uint32_t last_span = 1;
                KCHECK(expand_to_last_level(b, &last, &last_span, stream));
                uint64_t distinct = 0;
                int rc = KATOME_OK;
                {
                    PhaseScope ps(b->prof, PH_EXPAND_TILES, stream);
                    uint64_t n_rest = 0;
                    KCHECK(rest_valid(b, &n_rest, stream));
                    rc = tiles_to_edges_sorted_seen(*last, b->s.k, last_span, b->rc, 2ull * (b->seen_read_len - b->s.k + 1), b->edge_key, raw_seq, &b->n_edges,
                                                    &distinct, stream, n_rest ? b->rest_k.as<u64>() : nullptr, n_rest);
                    if (rc != KATOME_OK && rc != KATOME_E_UNSUPPORTED) return rc;
                }
                if (rc == KATOME_OK) {
                    b->tiles.release(); b->tiles2.release();
                    b->tiles_ready = false; b->tiles2_ready = false;
                    rest_reset(b);
                    b->stat_kmers = distinct; b->stat_kmer_slots = 0;
                    counted_seen = true;
                } else {
                    b->n_edges = 0;               // (numbers that do not pack, or a group too large: the table counts, from the tiles that are still there)
                }
            }
        }
        if (!counted && !counted_seen) KCHECK(expand_tiles(b, stream));
        if (!counted && (b->table_ready || counted_seen)) {
            if (!counted_seen) {
                KCHECK(table_occupied(b->table, &b->stat_kmers, stream));
                b->stat_kmer_slots = b->table.cap;
                PhaseScope ps(b->prof, PH_EMIT_EDGES, stream);
                // (first-seen order: the threshold is applied after the numbering, as the reference's retain passes do)
                if (b->first_seen) KCHECK(table_emit_edges(b->table, b->s.k, b->rc, 0, b->edge_key, raw_w, &b->n_edges, stream, &raw_seq));
                else KCHECK(table_emit_edges(b->table, b->s.k, b->rc, b->prune_weight, b->edge_key, b->edge_weight, &b->n_edges, stream));
            }
            for (int i = 0; i < 2; ++i) { b->scratch_k[i].release(); b->scratch_w[i].release(); }
            b->table.release();                       // the table is spent; its memory serves the sort
            b->table_ready = false;
            PhaseScope ps(b->prof, PH_SORT_EDGES, stream);
            if (b->first_seen) {
                // sort (key, position) and bring weight and sequence number along through the positions
                if (b->n_edges >= (1ull << 32)) { set_error("first-seen order: more than 2^32 edges on one GPU"); return KATOME_E_UNSUPPORTED; }
                DevBuf idx(stream);
                KCHECK(idx.alloc((b->n_edges + 1) * 4));
                KCHECK(dev_iota(idx.as<u32>(), b->n_edges, stream));
                KCHECK(dev_sort_bufs(b->edge_key, &idx, b->n_edges, b->nw, 2 * b->s.k, stream));
                KCHECK(b->edge_weight.alloc((b->n_edges + 1) * 4, stream));
                KCHECK(b->edge_seq.alloc((b->n_edges + 1) * 8, stream));
                KCHECK(dev_gather_seq_weight(raw_seq.as<u64>(), idx.as<u32>(), b->n_edges, b->edge_seq.as<u64>(), b->edge_weight.as<u32>(), stream));
            } else {
                KCHECK(dev_sort_bufs(b->edge_key, &b->edge_weight, b->n_edges, b->nw, 2 * b->s.k, stream, true));
            }
        } else if (!counted) {
            KCHECK(b->edge_key.alloc(16, stream)); KCHECK(b->edge_weight.alloc(16, stream));
        }
        b->edges_ready = true;
    }
    if (d_edge_key) *d_edge_key = b->edge_key.as<u64>();
    if (d_edge_weight) *d_edge_weight = b->edge_weight.as<u32>();
    if (n_edges) *n_edges = b->n_edges;
    return KATOME_OK;
}

int katome_dev_finalize(katome_builder* b, katome_dev_graph* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (b && b->finalized) return out ? katome_dev_current_graph(b, out) : KATOME_OK;      // (also: a graph installed by katome_dist_gather)
    KCHECK(katome_dev_edges(b, nullptr, nullptr, nullptr, stream_));
    const uint64_t E = b->n_edges;
    const uint32_t nw = b->nw, k = b->s.k;
    // node set = every source and target (k-1)-mer (add_fasta_node, pt_graph.rs:142-154): read off the
    // sorted edge list (radix.hip, dev_node_ids)
    KCHECK(b->edge_src.alloc((E + 1) * 8, stream));
    KCHECK(b->edge_dst.alloc((E + 1) * 8, stream));
    DevBuf node_first(stream);                 // first-seen order: the nodes' first touches come out of the same merge
    uint64_t n_marked = 0;                     // ... and, for targets below this index, "this edge touches it first" as a mark in edge_dst
    {
        PhaseScope ps(b->prof, PH_NODE_SET, stream);
        const bool fs = b->first_seen && E && b->edge_seq.p;
        KCHECK(dev_node_ids(b->edge_key.as<u64>(), E, k, b->node_key, b->edge_src.as<u64>(), b->edge_dst.as<u64>(), &b->n_nodes, stream,
                            fs ? b->edge_seq.as<u64>() : nullptr, fs ? &node_first : nullptr, &n_marked));
    }
    u64* cand = b->node_key.as<u64>();
    if (b->first_seen && E) {
        // Re-number everything the way the reference's loop would have: edges by the sequence number of their first
        // insertion, nodes by the first insertion that touches them (source of a strand's first window: 2*seq;
        // target: 2*seq + 1).  petgraph hands out indices in exactly that order (pt_graph.rs:149,194).
        PhaseScope ps(b->prof, PH_FIRST_SEEN, stream);
        const bool trace = getenv("KATOME_TRACE_FINALIZE") != nullptr;
        auto t_last = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!trace) return;
            (void)hipStreamSynchronize(stream);
            const auto t = std::chrono::steady_clock::now();
            fprintf(stderr, "[finalize] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
            t_last = t;
        };
        const uint64_t N = b->n_nodes;
        if (N >= (1ull << 32)) { set_error("first-seen order: more than 2^32 nodes on one GPU"); return KATOME_E_UNSUPPORTED; }
        const uint64_t max_seq = b->direct_edges ? 2 * b->direct_edges + 2
                                 : b->var_seq_base ? 2 * b->var_seq_base + 2
                                                   : 2 * (b->reads_inserted + 1) * 2 * (uint64_t)(b->seen_read_len - k + 1) + 2;
        uint32_t bits = 1;
        while (bits < 64 && (max_seq >> bits)) ++bits;
        // (buffers are taken and given back one at a time: at C3 every one of them is 6-13 GB)
        DevBuf new_id(stream), eperm(stream), aos(stream);
        if (!node_first.p) {
            KCHECK(node_first.alloc((N + 1) * 8));
            KCHECK_HIP(hipMemsetAsync(node_first.p, 0xFF, N * 8, stream));
            KCHECK(dev_node_first(b->edge_src.as<u64>(), b->edge_dst.as<u64>(), b->edge_seq.as<u64>(), E, node_first.as<u64>(), stream));
        }
        lap("node_first");
        KCHECK(eperm.alloc((E + 1) * 4));
        if (!getenv("KATOME_SORT_NODES") && aos.alloc(E * 32 + 64) == KATOME_OK) {
            // No sort of the nodes: every node is introduced by exactly one edge (the one whose first insertion is the node's
            // first touch), so with the edges in sequence order the node indices are a running count (radix.hip).
            KCHECK(dev_pack_edges_intro(b->edge_key.as<u64>(), b->edge_weight.as<u32>(), b->edge_src.as<u64>(), b->edge_dst.as<u64>(),
                                        b->edge_seq.as<u64>(), node_first.as<u64>(), E, nw, aos.p, stream, n_marked));
            node_first.release();
            lap("pack + who introduces");
            KCHECK(dev_iota(eperm.as<u32>(), E, stream));
            KCHECK(dev_sort_bufs(b->edge_seq, &eperm, E, 1, bits, stream));       // eperm[new] = old; edge_seq now ascending
            lap("sort edges by seq");
            DevBuf cnt(stream), offs(stream), onode(stream);
            KCHECK(cnt.alloc((E + 1) * 4));
            KCHECK(dev_unpack_edges_intro(aos.p, eperm.as<u32>(), E, nw, b->edge_key.as<u64>(), b->edge_weight.as<u32>(), b->edge_src.as<u64>(),
                                          b->edge_dst.as<u64>(), cnt.as<u32>(), stream));
            aos.release(); eperm.release();
            lap("edges to seq order");
            KCHECK(offs.alloc((E + 2) * 8));
            KCHECK(dev_scan_counts(cnt.as<u32>(), E, offs.as<u64>(), stream));
            uint64_t introduced = 0;
            KCHECK_HIP(hipMemcpyAsync(&introduced, offs.as<u64>() + E, 8, hipMemcpyDeviceToHost, stream));
            KCHECK_HIP(hipStreamSynchronize(stream));
            lap("scan");
            if (introduced != N) { set_error("first-seen order: %llu nodes introduced, %llu nodes known", (unsigned long long)introduced, (unsigned long long)N); return KATOME_E_DEVICE; }
            cnt.release();
            DevBuf osrc(stream), odst(stream);
            KCHECK(new_id.alloc((N + 1) * 8));
            KCHECK(onode.alloc((N + 1) * 8 * nw));
            KCHECK(osrc.alloc((E + 1) * 8));
            KCHECK(odst.alloc((E + 1) * 8));
            KCHECK(dev_assign_nodes(b->edge_key.as<u64>(), b->edge_src.as<u64>(), b->edge_dst.as<u64>(), offs.as<u64>(), E, nw, k,
                                    new_id.as<u64>(), onode.as<u64>(), osrc.as<u64>(), odst.as<u64>(), stream));
            { const size_t n = onode.bytes; b->node_key.adopt(onode.take(), n); }
            { const size_t n = osrc.bytes; b->edge_src.adopt(osrc.take(), n); }
            { const size_t n = odst.bytes; b->edge_dst.adopt(odst.take(), n); }
            lap("node indices + end points");
        } else {
        aos.release();
        KCHECK(dev_clear_dst_marks(b->edge_dst.as<u64>(), E, stream));      // (the merge's marks: only the other branch reads them)
        KCHECK(new_id.alloc((N + 1) * 8));
        {
            DevBuf nperm(stream), onode(stream);
            KCHECK(nperm.alloc((N + 1) * 4));
            KCHECK(dev_iota(nperm.as<u32>(), N, stream));
            KCHECK(dev_sort_bufs(node_first, &nperm, N, 1, bits, stream));        // nperm[new] = old
            lap("sort nodes");
            node_first.release();
            KCHECK(dev_invert(nperm.as<u32>(), N, new_id.as<u64>(), stream));                   // new_id[old] = new
            KCHECK(onode.alloc((N + 1) * 8 * nw));
            KCHECK(dev_gather_keys(b->node_key.as<u64>(), nperm.as<u32>(), N, nw, onode.as<u64>(), stream));
            const size_t n = onode.bytes; b->node_key.adopt(onode.take(), n);
            lap("invert + node keys");
        }
        KCHECK(dev_iota(eperm.as<u32>(), E, stream));
        KCHECK(dev_sort_bufs(b->edge_seq, &eperm, E, 1, bits, stream));           // eperm[new] = old; edge_seq now ascending
        lap("sort edges by seq");
        {
            // one 32-byte record per edge, read once at random (radix.hip dev_permute_edges); if that much scratch is not
            // to be had, the four separate gathers
            DevBuf aos2(stream);
            if (aos2.alloc(E * 32 + 64) == KATOME_OK) {
                KCHECK(dev_permute_edges(b->edge_key.as<u64>(), b->edge_weight.as<u32>(), b->edge_src.as<u64>(), b->edge_dst.as<u64>(),
                                         new_id.as<u64>(), eperm.as<u32>(), E, nw, aos2.p, stream));
            } else {
                {
                    DevBuf o(stream);
                    KCHECK(o.alloc((E + 1) * 8 * nw));
                    KCHECK(dev_gather_keys(b->edge_key.as<u64>(), eperm.as<u32>(), E, nw, o.as<u64>(), stream));
                    const size_t n = o.bytes; b->edge_key.adopt(o.take(), n);
                }
                {
                    DevBuf o(stream);
                    KCHECK(o.alloc((E + 1) * 4));
                    KCHECK(dev_gather_u32(b->edge_weight.as<u32>(), eperm.as<u32>(), E, o.as<u32>(), stream));
                    const size_t n = o.bytes; b->edge_weight.adopt(o.take(), n);
                }
                {
                    DevBuf o(stream);
                    KCHECK(o.alloc((E + 1) * 8));
                    KCHECK(dev_gather_mapped(b->edge_src.as<u64>(), eperm.as<u32>(), new_id.as<u64>(), E, o.as<u64>(), stream));
                    const size_t n = o.bytes; b->edge_src.adopt(o.take(), n);
                }
                {
                    DevBuf o(stream);
                    KCHECK(o.alloc((E + 1) * 8));
                    KCHECK(dev_gather_mapped(b->edge_dst.as<u64>(), eperm.as<u32>(), new_id.as<u64>(), E, o.as<u64>(), stream));
                    const size_t n = o.bytes; b->edge_dst.adopt(o.take(), n);
                }
            }
        }
        }
        lap("edges to seq order");
        if (b->prune_weight) KCHECK(weak_edges_ordered(b, b->prune_weight, stream));
        cand = b->node_key.as<u64>();
    }
    const uint32_t stride = label_stride_for_k(k);
    KCHECK(b->edge_label.alloc((E + 1) * (size_t)stride + 16, stream));
    {
        PhaseScope ps(b->prof, PH_LABELS, stream);
        KCHECK(dev_labels(b->edge_key.as<u64>(), b->n_edges, k, b->edge_label.as<uint8_t>(), stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    b->finalized = true;
    if (out) {
        out->n_nodes = b->n_nodes; out->n_edges = b->n_edges;
        out->key_words = nw; out->label_stride = stride;
        out->d_edge_key = b->edge_key.as<u64>(); out->d_edge_weight = b->edge_weight.as<u32>();
        out->d_edge_src = b->edge_src.as<u64>(); out->d_edge_dst = b->edge_dst.as<u64>();
        out->d_edge_label = b->edge_label.as<uint8_t>(); out->d_node_key = cand;
        out->d_edge_age = b->edge_age.p ? b->edge_age.as<u32>() : nullptr;
    }
    return KATOME_OK;
}

int katome_dev_remove_dead_paths(katome_builder* b, katome_dev_graph* out, katome_prune_stats* stats, void* stream_) {
    if (!b) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (!b->first_seen) { set_error("remove_dead_paths needs a KATOME_FLAG_FIRST_SEEN_ORDER builder"); return KATOME_E_ARG; }
    if (!b->finalized) { set_error("remove_dead_paths: call katome_dev_finalize first"); return KATOME_E_ARG; }
    const auto t0 = std::chrono::steady_clock::now();
    PruneGraph g{&b->edge_src, &b->edge_dst, &b->edge_weight, &b->edge_key, &b->node_key, &b->edge_age, b->n_edges, b->n_nodes, b->nw};
    g.parallel_edges = b->direct_edges != 0;
    katome_prune_stats st;
    {
        PhaseScope ps(b->prof, PH_DEAD_PATHS, stream);
        KCHECK(dev_remove_dead_paths(g, b->s.k, &st, stream));
        b->n_edges = g.n_edges; b->n_nodes = g.n_nodes;
        // the labels follow their edges: rewrite them from the keys (kmer_to_edge, compress.rs:231-233)
        KCHECK(dev_labels(b->edge_key.as<u64>(), b->n_edges, b->s.k, b->edge_label.as<uint8_t>(), stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    st.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (stats) *stats = st;
    if (out) {
        out->n_nodes = b->n_nodes; out->n_edges = b->n_edges;
        out->key_words = b->nw; out->label_stride = label_stride_for_k(b->s.k);
        out->d_edge_key = b->edge_key.as<u64>(); out->d_edge_weight = b->edge_weight.as<u32>();
        out->d_edge_src = b->edge_src.as<u64>(); out->d_edge_dst = b->edge_dst.as<u64>();
        out->d_edge_label = b->edge_label.as<uint8_t>(); out->d_node_key = b->node_key.as<u64>();
        out->d_edge_age = b->edge_age.p ? b->edge_age.as<u32>() : nullptr;
    }
    return KATOME_OK;
}

int katome_dev_standardize_contigs(katome_builder* b, void* stream_) {
    if (!b) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (!b->finalized) { set_error("standardize_contigs: call katome_dev_finalize first"); return KATOME_E_ARG; }
    KCHECK(dev_standardize_contigs(b->edge_src.as<u64>(), b->edge_dst.as<u64>(), b->edge_weight.as<u32>(), b->n_edges, b->n_nodes, stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    return KATOME_OK;
}

int katome_dev_standardize_edges(katome_builder* b, uint64_t original_genome_length, uint32_t threshold, void* stream_) {
    if (!b) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (!b->finalized) { set_error("standardize_edges: call katome_dev_finalize first"); return KATOME_E_ARG; }
    KCHECK(dev_standardize_scale(b->edge_weight.as<u32>(), b->n_edges, original_genome_length, b->s.k, threshold, stream));
    KCHECK(weak_edges_ordered(b, 1, stream));          // "remove edges with weight 0" (standardizer.rs:68-69)
    KCHECK(dev_labels(b->edge_key.as<u64>(), b->n_edges, b->s.k, b->edge_label.as<uint8_t>(), stream));
    KCHECK_HIP(hipStreamSynchronize(stream));
    return KATOME_OK;
}

int katome_dev_shrink(katome_builder* b, katome_dev_contigs* out, void* stream_) {
    return katome_dev_shrink_mode(b, KATOME_SHRINK_AUTO, out, nullptr, stream_);
}
int katome_dev_shrink_mode(katome_builder* b, uint32_t mode, katome_dev_contigs* out, double* host_ms, void* stream_) {
    if (!b || !out) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK_HIP(hipSetDevice(b->s.device));
    if (host_ms) *host_ms = 0;
    if (!b->finalized) { set_error("shrink: call katome_dev_finalize first"); return KATOME_E_ARG; }
    if (mode > KATOME_SHRINK_EXACT) { set_error("shrink: unknown mode %u", mode); return KATOME_E_ARG; }
    // (auto: a graph in the reference's numbering gets the reference's own result -- its numbering is what the stage after it reads)
    const bool exact = mode == KATOME_SHRINK_EXACT || (mode == KATOME_SHRINK_AUTO && b->first_seen);
    ShrinkInput in{b->edge_src.as<u64>(), b->edge_dst.as<u64>(), b->edge_weight.as<u32>(), b->edge_key.as<u64>(), b->node_key.as<u64>(),
                   b->n_edges, b->n_nodes, b->nw, b->s.k};
    {
        PhaseScope ps(b->prof, PH_SHRINK, stream);
        if (exact) KCHECK(dev_shrink_exact(in, b->edge_age.p ? b->edge_age.as<u32>() : nullptr, b->shrunk, host_ms, stream));
        else KCHECK(dev_shrink(in, b->shrunk, stream));
    }
    memset(out, 0, sizeof *out);
    out->n_nodes = b->shrunk.n_nodes; out->n_edges = b->shrunk.n_edges; out->label_bytes = b->shrunk.label_bytes;
    out->key_words = b->nw;
    out->d_edge_src = b->shrunk.edge_src.as<u64>(); out->d_edge_dst = b->shrunk.edge_dst.as<u64>();
    out->d_edge_weight = b->shrunk.edge_weight.as<u32>(); out->d_edge_kmers = b->shrunk.edge_kmers.as<u32>();
    out->d_edge_label_off = b->shrunk.edge_label_off.as<u64>(); out->d_edge_label = b->shrunk.edge_label.as<uint8_t>();
    out->d_node_key = b->shrunk.node_key.as<u64>();
    return KATOME_OK;
}

int katome_dev_current_graph(katome_builder* b, katome_dev_graph* out) {
    if (!b || !out) { set_error("null argument"); return KATOME_E_ARG; }
    if (!b->finalized) { set_error("katome_dev_current_graph: call katome_dev_finalize first"); return KATOME_E_ARG; }
    out->n_nodes = b->n_nodes; out->n_edges = b->n_edges;
    out->key_words = b->nw; out->label_stride = label_stride_for_k(b->s.k);
    out->d_edge_key = b->edge_key.as<u64>(); out->d_edge_weight = b->edge_weight.as<u32>();
    out->d_edge_src = b->edge_src.as<u64>(); out->d_edge_dst = b->edge_dst.as<u64>();
    out->d_edge_label = b->edge_label.as<uint8_t>(); out->d_node_key = b->node_key.as<u64>();
        out->d_edge_age = b->edge_age.p ? b->edge_age.as<u32>() : nullptr;
    return KATOME_OK;
}

int katome_builder_counts(katome_builder* b, uint64_t* out8) {
    if (!b || !out8) { set_error("null argument"); return KATOME_E_ARG; }
    out8[0] = b->stat_tiles; out8[1] = b->stat_tile_slots; out8[2] = b->stat_kmers; out8[3] = b->stat_kmer_slots;
    out8[4] = b->stat_tiles2; out8[5] = b->stat_tile2_slots; out8[6] = b->span; out8[7] = b->span2;
    return KATOME_OK;
}

int katome_builder_profile(katome_builder* b, int enable) {
    if (!b) { set_error("null argument"); return KATOME_E_ARG; }
    b->prof.on = enable != 0;
    b->prof.clear();
    return KATOME_OK;
}
uint32_t katome_phase_count(void) { return PH_COUNT; }
const char* katome_phase_name(uint32_t phase) { return phase < PH_COUNT ? PHASE_NAMES[phase] : ""; }
int katome_builder_profile_read_work(katome_builder* b, double* total_ms, uint64_t* launches, uint64_t* work) {
    if (!b || !total_ms || !launches) { set_error("null argument"); return KATOME_E_ARG; }
    KCHECK_HIP(hipSetDevice(b->s.device));
    KCHECK_HIP(hipDeviceSynchronize());
    for (int i = 0; i < PH_COUNT; ++i) { total_ms[i] = 0; launches[i] = 0; if (work) work[i] = 0; }
    for (auto& e : b->prof.evs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { total_ms[e.phase] += ms; launches[e.phase] += 1; if (work) work[e.phase] += e.work; }
    }
    b->prof.clear();
    return KATOME_OK;
}
int katome_builder_profile_read(katome_builder* b, double* total_ms, uint64_t* launches) {
    return katome_builder_profile_read_work(b, total_ms, launches, nullptr);
}

int katome_dev_release_cache(int device) {
    KCHECK(use_device(device));
    dev_release_cache(device);
    return KATOME_OK;
}

int katome_dev_cache_stats(int device, uint64_t out[3]) {
    if (!out) { set_error("null argument"); return KATOME_E_ARG; }
    dev_cache_stats(device, out);
    return KATOME_OK;
}

int katome_dev_sort(int device, uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t key_words, uint32_t key_bits, void* stream) {
    KCHECK(use_device(device));
    return dev_sort(d_keys, d_vals, n, key_words, key_bits, (hipStream_t)stream);
}
int katome_dev_replay_edge_removals(int device, const uint32_t* d_pos, const uint32_t* d_mult, uint64_t u, uint64_t n_edges,
                                    uint32_t* d_victims, uint32_t* d_move_to, uint32_t* d_move_from, uint64_t* counts, void* stream_) {
    KCHECK(use_device(device));
    if (!counts || (u && (!d_pos || !d_mult || !d_victims || !d_move_to || !d_move_from))) { set_error("null argument"); return KATOME_E_ARG; }
    if (n_edges >= 0xFFFFFFFFull) { set_error("more than 2^32 edges"); return KATOME_E_UNSUPPORTED; }
    hipStream_t stream = (hipStream_t)stream_;
    ReplayScratch sc(stream);
    DevBuf victims(stream), to(stream), from(stream);
    uint64_t m = 0, left = n_edges, moves = 0, dups = 0;
    KCHECK(dev_replay_edges(d_pos, d_mult, u, n_edges, sc, victims, to, from, &m, &left, &moves, &dups, stream));
    if (m) KCHECK_HIP(hipMemcpyAsync(d_victims, victims.p, m * 4, hipMemcpyDeviceToDevice, stream));
    if (moves) {
        KCHECK_HIP(hipMemcpyAsync(d_move_to, to.p, moves * 4, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(d_move_from, from.p, moves * 4, hipMemcpyDeviceToDevice, stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    counts[0] = m; counts[1] = moves; counts[2] = left; counts[3] = dups;
    return KATOME_OK;
}
int katome_dev_replay_node_removals(int device, const uint32_t* d_die, uint64_t m, uint64_t n_nodes, uint32_t* d_move_to, uint32_t* d_move_from,
                                    uint64_t* counts, void* stream_) {
    KCHECK(use_device(device));
    if (!counts || (m && (!d_die || !d_move_to || !d_move_from))) { set_error("null argument"); return KATOME_E_ARG; }
    if (n_nodes >= 0xFFFFFFFFull || m >= 0x7FFFFFFFull) { set_error("more than 2^32 nodes"); return KATOME_E_UNSUPPORTED; }
    hipStream_t stream = (hipStream_t)stream_;
    NodeReplayScratch sc(stream);
    DevBuf to(stream), from(stream);
    uint64_t moves = 0, left = n_nodes;
    int fell_back = 0;
    KCHECK(dev_replay_nodes(d_die, m, n_nodes, sc, to, from, &moves, &left, &fell_back, stream));
    if (moves) {
        KCHECK_HIP(hipMemcpyAsync(d_move_to, to.p, moves * 4, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(d_move_from, from.p, moves * 4, hipMemcpyDeviceToDevice, stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    counts[0] = moves; counts[1] = left; counts[2] = (uint64_t)fell_back;
    return KATOME_OK;
}
// the same two replays on 64-bit positions (dist_prune.hip: a graph sharded over several GPUs may hold more than 2^32 edges);
// only the marked entries and the tail that disappears are touched, so n_edges / n_nodes may be far beyond any buffer here
int katome_dev_replay_edge_removals64(int device, const uint64_t* d_pos, const uint32_t* d_mult, uint64_t u, uint64_t n_edges,
                                      uint64_t* d_victims, uint64_t* d_move_to, uint64_t* d_move_from, uint64_t* counts, void* stream_) {
    KCHECK(use_device(device));
    if (!counts || (u && (!d_pos || !d_mult || !d_victims || !d_move_to || !d_move_from))) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    ReplayScratch sc(stream);
    DevBuf victims(stream), to(stream), from(stream);
    uint64_t m = 0, left = n_edges, moves = 0, dups = 0;
    KCHECK(dev_replay_edges64(d_pos, d_mult, u, n_edges, sc, victims, to, from, &m, &left, &moves, &dups, stream));
    if (m) KCHECK_HIP(hipMemcpyAsync(d_victims, victims.p, m * 8, hipMemcpyDeviceToDevice, stream));
    if (moves) {
        KCHECK_HIP(hipMemcpyAsync(d_move_to, to.p, moves * 8, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(d_move_from, from.p, moves * 8, hipMemcpyDeviceToDevice, stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    counts[0] = m; counts[1] = moves; counts[2] = left; counts[3] = dups;
    return KATOME_OK;
}
int katome_dev_replay_node_removals64(int device, const uint64_t* d_die, uint64_t m, uint64_t n_nodes, uint64_t* d_move_to, uint64_t* d_move_from,
                                      uint64_t* counts, void* stream_) {
    KCHECK(use_device(device));
    if (!counts || (m && (!d_die || !d_move_to || !d_move_from))) { set_error("null argument"); return KATOME_E_ARG; }
    hipStream_t stream = (hipStream_t)stream_;
    NodeReplayScratch sc(stream);
    DevBuf to(stream), from(stream);
    uint64_t moves = 0, left = n_nodes;
    int fell_back = 0;
    KCHECK(dev_replay_nodes64(d_die, m, n_nodes, sc, to, from, &moves, &left, &fell_back, stream));
    if (moves) {
        KCHECK_HIP(hipMemcpyAsync(d_move_to, to.p, moves * 8, hipMemcpyDeviceToDevice, stream));
        KCHECK_HIP(hipMemcpyAsync(d_move_from, from.p, moves * 8, hipMemcpyDeviceToDevice, stream));
    }
    KCHECK_HIP(hipStreamSynchronize(stream));
    counts[0] = moves; counts[1] = left; counts[2] = (uint64_t)fell_back;
    return KATOME_OK;
}
int katome_dev_scan_counts(int device, const uint32_t* d_counts, uint64_t m, uint64_t* d_offs, void* stream) {
    KCHECK(use_device(device));
    if (!d_offs || (m && !d_counts)) { set_error("null argument"); return KATOME_E_ARG; }
    return dev_scan_counts(d_counts, m, d_offs, (hipStream_t)stream);
}
int katome_dev_unique(int device, uint64_t* d_keys, uint64_t n, uint32_t key_words, uint64_t* n_out, void* stream) {
    KCHECK(use_device(device));
    if (key_words != 1 && key_words != 2) { set_error("key_words must be 1 or 2"); return KATOME_E_ARG; }
    return dev_unique(d_keys, n, key_words, n_out, (hipStream_t)stream);
}
int katome_dev_rank(int device, const uint64_t* d_sorted, uint64_t n_sorted, uint32_t key_words, uint32_t key_bits,
                    const uint64_t* d_queries, uint64_t n_queries, uint64_t* d_rank_out, void* stream) {
    KCHECK(use_device(device));
    if (key_words != 1 && key_words != 2) { set_error("key_words must be 1 or 2"); return KATOME_E_ARG; }
    return dev_rank(d_sorted, n_sorted, key_words, key_bits, d_queries, n_queries, d_rank_out, (hipStream_t)stream);
}
int katome_dev_node_ids(int device, const uint64_t* d_edge_key, uint64_t n_edges, uint32_t k, uint64_t* d_node_key,
                        uint64_t* d_edge_src, uint64_t* d_edge_dst, uint64_t* n_nodes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK(use_device(device));
    KCHECK(check_k(k));
    DevBuf nodes(stream);
    KCHECK(dev_node_ids(d_edge_key, n_edges, k, nodes, d_edge_src, d_edge_dst, n_nodes, stream));
    if (*n_nodes) KCHECK_HIP(hipMemcpyAsync(d_node_key, nodes.p, *n_nodes * 8 * key_words_for_k(k), hipMemcpyDeviceToDevice, stream));
    return KATOME_OK;
}

int katome_dev_source_ids(int device, const uint64_t* d_edge_key, uint64_t n_edges, uint32_t k, uint64_t* d_node_key,
                          uint64_t* d_edge_src, uint64_t* n_sources, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    KCHECK(use_device(device));
    KCHECK(check_k(k));
    DevBuf nodes(stream);
    KCHECK(dev_source_ids(d_edge_key, n_edges, k, nodes, d_edge_src, n_sources, stream));
    if (*n_sources) KCHECK_HIP(hipMemcpyAsync(d_node_key, nodes.p, *n_sources * 8 * key_words_for_k(k), hipMemcpyDeviceToDevice, stream));
    return KATOME_OK;
}
int katome_dev_endpoints(int device, const uint64_t* d_edge_key, uint64_t n, uint32_t k, uint64_t* d_src_key, uint64_t* d_dst_key, void* stream) {
    KCHECK(use_device(device));
    KCHECK(check_k(k));
    return dev_endpoints(d_edge_key, n, k, d_src_key, d_dst_key, (hipStream_t)stream);
}
int katome_dev_labels(int device, const uint64_t* d_edge_key, uint64_t n, uint32_t k, uint8_t* d_label, void* stream) {
    KCHECK(use_device(device));
    KCHECK(check_k(k));
    return dev_labels(d_edge_key, n, k, d_label, (hipStream_t)stream);
}
int katome_dev_synth_reads(int device, uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len,
                           double err_rate, uint32_t n_inject_percent, uint8_t* d_packed, uint8_t* d_skip, void* stream) {
    KCHECK(use_device(device));
    return launch_synth(first_read, n_reads, read_len, genome_len, err_rate, n_inject_percent, d_packed, d_skip, (hipStream_t)stream);
}

// ---- host-memory entry points ---------------------------------------------------------------------
struct GraphOwner {            // katome_graph followed by what it owns
    katome_graph g;
    std::vector<void*> mem;
};

void katome_graph_free(katome_graph* g) {
    if (!g) return;
    GraphOwner* o = reinterpret_cast<GraphOwner*>(g);
    for (void* p : o->mem) free(p);
    delete o;
}

struct ContigsOwner {          // katome_contigs followed by what it owns
    katome_contigs c;
    std::vector<void*> mem;
};
void katome_contigs_free(katome_contigs* c) {
    if (!c) return;
    ContigsOwner* o = reinterpret_cast<ContigsOwner*>(c);
    for (void* p : o->mem) free(p);
    delete o;
}

}  // extern "C"

// KATOME_TRACE_BUILD=1: wall time of the host entries' stages on stderr
static void build_lap(const char* what, bool reset = false) {
    static const bool on = getenv("KATOME_TRACE_BUILD") != nullptr;
    static std::chrono::steady_clock::time_point last;
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    if (!reset) fprintf(stderr, "[build] %-28s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
}

// Host memory for a result array.  A device -> host copy into pages the process has never touched runs at the rate one
// thread takes page faults (~9 GB/s measured on the MI355X box; into touched pages, pinned or not, the same copy runs at
// ~55 GB/s), so large arrays are taken 2 MiB-aligned, offered to the kernel as huge pages and first touched by several
// threads at once.  Released with free().
static void* host_result_reserve(size_t bytes) {          // (pages not touched yet)
    const size_t big = (size_t)64 << 20, huge = (size_t)2 << 20;
    if (bytes < big) return malloc(std::max<size_t>(bytes, 1));
    void* p = nullptr;
    if (posix_memalign(&p, huge, (bytes + huge - 1) / huge * huge) != 0) return nullptr;
    (void)madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}
static void host_result_touch(void* p, size_t bytes) {
    if (bytes < ((size_t)64 << 20)) return;
    unsigned T = std::thread::hardware_concurrency();
    T = std::max(1u, std::min(T ? T : 4u, 16u));
    const size_t pages = (bytes + 4095) / 4096, per = (pages + T - 1) / T;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back([=]() {
            volatile char* c = static_cast<volatile char*>(p);
            for (size_t pg = t * per; pg < std::min(pages, (t + 1) * per); ++pg) c[pg * 4096] = 0;
        });
    for (auto& x : th) x.join();
}
static void* host_result_alloc(size_t bytes) {
    void* p = host_result_reserve(bytes);
    if (p) host_result_touch(p, bytes);
    return p;
}

// Several result arrays: while array i comes over PCIe, the pages of array i + 1 are being touched (a third of the time of
// a 7 GB graph was the touching, done array by array in front of each copy).
struct HostCopy { void** dst; const void* src; size_t bytes; void* h = nullptr; };
template <class Owner> static int d2h_all(Owner* o, std::vector<HostCopy>& jobs) {
    for (auto& j : jobs) {
        j.h = host_result_reserve(j.bytes);
        if (!j.h) { set_error("out of host memory"); return KATOME_E_OOM; }
        o->mem.push_back(j.h);
        *j.dst = j.h;
    }
    std::mutex m; std::condition_variable cv; size_t touched = 0;
    std::thread toucher([&]() {
        for (auto& j : jobs) {
            host_result_touch(j.h, j.bytes);
            { std::lock_guard<std::mutex> lk(m); ++touched; }
            cv.notify_all();
        }
    });
    int rc = KATOME_OK;
    for (size_t i = 0; i < jobs.size(); ++i) {
        { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&]() { return touched > i; }); }
        if (rc == KATOME_OK && jobs[i].bytes && hipMemcpy(jobs[i].h, jobs[i].src, jobs[i].bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
            rc = KATOME_E_DEVICE;
        }
    }
    toucher.join();
    return rc;
}

template <class T, class Owner> static int d2h(Owner* o, const T** dst, const void* d_src, size_t count) {
    T* h = (T*)host_result_alloc(std::max<size_t>(count, 1) * sizeof(T));
    if (!h) { set_error("out of host memory"); return KATOME_E_OOM; }
    o->mem.push_back(h);
    if (count) KCHECK_HIP(hipMemcpy(h, d_src, count * sizeof(T), hipMemcpyDeviceToHost));
    *dst = h;
    return KATOME_OK;
}

// the stages of assemble_with_graph (asm/basic_assembler.rs:58-72) a host entry may ask for after the build, in the
// order given: d = remove_dead_paths, c = standardize_contigs, w = remove_weak_edges(min_weight),
// e = standardize_edges(original_genome_length, k, min_weight)
static int run_stages(katome_builder* b, const char* stages, uint64_t genome_len) {
    for (const char* st = stages ? stages : ""; *st; ++st) {
        switch (*st) {
            case 'd': KCHECK(katome_dev_remove_dead_paths(b, nullptr, nullptr, nullptr)); break;
            case 'c': KCHECK(katome_dev_standardize_contigs(b, nullptr)); break;
            case 'w': KCHECK(katome_dev_remove_weak_edges(b, b->s.min_weight, nullptr)); break;
            case 'e': KCHECK(katome_dev_standardize_edges(b, genome_len, b->s.min_weight, nullptr)); break;
            default: set_error("unknown stage '%c' (d, c, w, e)", *st); return KATOME_E_ARG;
        }
    }
    return KATOME_OK;
}

static int graph_to_host(katome_builder* b, uint64_t read_bytes, katome_graph** out, const char* stages = nullptr, uint64_t genome_len = 0) {
    katome_dev_graph dg;
    if (stages && *stages && !b->first_seen) { set_error("stages after the build need KATOME_FLAG_FIRST_SEEN_ORDER"); return KATOME_E_ARG; }
    build_lap("counting (H2D, kernels)");
    KCHECK(katome_dev_finalize(b, &dg, nullptr));
    build_lap("finalize");
    if (b->s.flags & KATOME_FLAG_REMOVE_DEAD_PATHS) KCHECK(katome_dev_remove_dead_paths(b, &dg, nullptr, nullptr));
    KCHECK(run_stages(b, stages, genome_len));
    KCHECK(katome_dev_current_graph(b, &dg));
    build_lap("stages after the build");
    GraphOwner* o = new (std::nothrow) GraphOwner();
    if (!o) { set_error("out of host memory"); return KATOME_E_OOM; }
    memset(&o->g, 0, sizeof o->g);
    katome_graph* g = &o->g;
    g->n_nodes = dg.n_nodes; g->n_edges = dg.n_edges; g->read_bytes = read_bytes;
    g->k = b->s.k; g->key_words = dg.key_words; g->label_stride = dg.label_stride;
    std::vector<HostCopy> jobs = {
        {(void**)&g->edge_src, dg.d_edge_src, dg.n_edges * 8}, {(void**)&g->edge_dst, dg.d_edge_dst, dg.n_edges * 8},
        {(void**)&g->edge_weight, dg.d_edge_weight, dg.n_edges * 4},
        {(void**)&g->edge_label, dg.d_edge_label, dg.n_edges * (size_t)dg.label_stride},
        {(void**)&g->edge_key, dg.d_edge_key, dg.n_edges * 8 * (size_t)dg.key_words},
        {(void**)&g->node_key, dg.d_node_key, dg.n_nodes * 8 * (size_t)dg.key_words}};
    if (dg.d_edge_age) jobs.push_back({(void**)&g->edge_age, dg.d_edge_age, dg.n_edges * 4});
    const int rc = d2h_all(o, jobs);
    if (rc) { katome_graph_free(g); return rc; }
    build_lap("graph to host arrays");
    *out = g;
    return KATOME_OK;
}

// finalize (+ the pruning the flags ask for) + shrink, copied to host arrays
static int contigs_to_host(katome_builder* b, uint64_t read_bytes, katome_contigs** out) {
    katome_dev_graph dg;
    KCHECK(katome_dev_finalize(b, &dg, nullptr));
    if (b->s.flags & KATOME_FLAG_REMOVE_DEAD_PATHS) KCHECK(katome_dev_remove_dead_paths(b, &dg, nullptr, nullptr));
    katome_dev_contigs dc;
    KCHECK(katome_dev_shrink(b, &dc, nullptr));
    ContigsOwner* o = new (std::nothrow) ContigsOwner();
    if (!o) { set_error("out of host memory"); return KATOME_E_OOM; }
    memset(&o->c, 0, sizeof o->c);
    katome_contigs* c = &o->c;
    c->n_nodes = dc.n_nodes; c->n_edges = dc.n_edges; c->label_bytes = dc.label_bytes; c->read_bytes = read_bytes;
    c->k = b->s.k; c->key_words = dc.key_words;
    int rc = KATOME_OK;
    if ((rc = d2h(o, &c->edge_src, dc.d_edge_src, dc.n_edges)) || (rc = d2h(o, &c->edge_dst, dc.d_edge_dst, dc.n_edges)) ||
        (rc = d2h(o, &c->edge_weight, dc.d_edge_weight, dc.n_edges)) || (rc = d2h(o, &c->edge_kmers, dc.d_edge_kmers, dc.n_edges)) ||
        (rc = d2h(o, &c->edge_label_off, dc.d_edge_label_off, dc.n_edges ? dc.n_edges + 1 : 0)) ||
        (rc = d2h(o, &c->edge_label, dc.d_edge_label, dc.label_bytes)) ||
        (rc = d2h(o, &c->node_key, dc.d_node_key, dc.n_nodes * dc.key_words))) {
        katome_contigs_free(c);
        return rc;
    }
    if (dc.n_edges == 0) const_cast<uint64_t*>(c->edge_label_off)[0] = 0;       // (d2h hands out room for one entry even when asked for none)
    *out = c;
    return KATOME_OK;
}

// what a host entry hands back: the graph, or the graph after shrink
struct Finish {
    katome_graph** graph; katome_contigs** contigs;
    const char* stages = nullptr; uint64_t genome_len = 0;
    int operator()(katome_builder* b, uint64_t read_bytes) const {
        return contigs ? contigs_to_host(b, read_bytes, contigs) : graph_to_host(b, read_bytes, graph, stages, genome_len);
    }
};

// records per extraction batch: bounded by a slice of free device memory
static uint64_t batch_records(uint32_t nw) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 1ull << 24;
    free_b += dev_cached_bytes();
    uint64_t r = (uint64_t)(free_b / 8) / (8ull * nw);
    return std::min<uint64_t>(std::max<uint64_t>(r, 1ull << 20), 1ull << 30);
}

extern "C" {

}  // extern "C"

// settings.n_devices > 1: the sharded build (dist.hip) with the ranks as host threads of this call, one per GPU.  Reads are
// split contiguously by index; every rank copies its own share to its GPU.  By packed key the host arrays are the ranks'
// shares one after the other (every rank copies its slice out itself); in the reference's numbering the graph is
// gathered to the first GPU in index order, where the stages the flags ask for run and the result is copied out.
static int build_packed_multi(const katome_settings* s, const uint8_t* packed, uint64_t n_reads, uint32_t read_len,
                              const uint8_t* skip, const Finish& finish, uint64_t read_bytes) {
    const int n = s->n_devices;
    const bool share = (s->flags & KATOME_FLAG_RANKS_SHARE_DEVICE) != 0, first_seen = (s->flags & KATOME_FLAG_FIRST_SEEN_ORDER) != 0;
    if (n > KATOME_MAX_RANKS) { set_error("n_devices = %d: at most %d", n, KATOME_MAX_RANKS); return KATOME_E_UNSUPPORTED; }
    KCHECK(use_device(s->device));
    int n_visible = 0;
    KCHECK_HIP(hipGetDeviceCount(&n_visible));
    if (!share && s->device + n > n_visible) { set_error("n_devices = %d from device %d, but %d GPU(s) are visible", n, s->device, n_visible); return KATOME_E_DEVICE; }
    if ((finish.contigs || (finish.stages && *finish.stages) || (s->flags & KATOME_FLAG_REMOVE_DEAD_PATHS)) && !first_seen) {
        set_error("n_devices > 1: shrink and the stages after the build need KATOME_FLAG_FIRST_SEEN_ORDER (they run on the graph gathered in the reference's numbering)");
        return KATOME_E_ARG;
    }
    // first-seen order: shrink and the staged pipeline still run on the graph gathered to the first GPU; the build itself and
    // remove_dead_paths do not (KATOME_DIST_PRUNE=gather: the gathered route for those too)
    const char* prune_route = getenv("KATOME_DIST_PRUNE");
    const bool direct = first_seen && !finish.contigs && !(finish.stages && *finish.stages) && !(prune_route && !strcmp(prune_route, "gather"));
    std::vector<int> devices(n);
    for (int r = 0; r < n; ++r) devices[r] = share ? s->device : s->device + r;
    std::vector<katome_comm*> comms(n, nullptr);
    const char* transport = getenv("KATOME_COMM");
    auto sync = std::make_shared<LocalGroup>(n);                 // host rendezvous of the rank threads, whatever moves the data
    std::shared_ptr<LocalGroup> group;                           // (the local transport's own rendezvous, poisoned with `sync`)
    if (share || (transport && !strcmp(transport, "local"))) {
        group = std::make_shared<LocalGroup>(n);
        for (int r = 0; r < n; ++r) KCHECK(make_local_comm(group, r, devices[r], &comms[r]));
    } else {
        KCHECK(make_rccl_comms_all(devices.data(), n, comms.data()));
    }
    struct Shared {
        std::vector<int> rc; std::vector<std::string> err;
        std::vector<uint64_t> n_edges, node_base, n_nodes;
        GraphOwner* owner = nullptr; int alloc_rc = KATOME_OK;
        uint64_t total_edges = 0, total_nodes = 0;
    } sh;
    sh.rc.assign(n, KATOME_OK); sh.err.resize(n); sh.n_edges.assign(n, 0); sh.node_base.assign(n, 0); sh.n_nodes.assign(n, 0);
    const uint32_t stride = (read_len + 3) / 4;
    auto body = [&](int r) -> int {
        KCHECK(use_device(devices[r]));
        hipStream_t stream = nullptr;
        KCHECK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        katome_settings mine = *s;
        mine.device = devices[r];
        katome_dist_builder* d = nullptr;
        int rc = katome_dist_create(&mine, comms[r], &d);
        do {
            if (rc) break;
            uint64_t first = 0, cnt = 0;
            katome_shard_range(n_reads, (uint32_t)n, (uint32_t)r, &first, &cnt);
            DevBuf d_packed(stream), d_skip(stream);
            if ((rc = d_packed.alloc(cnt * stride + 32))) break;
            if (cnt && hipMemcpyAsync(d_packed.p, packed + first * stride, cnt * stride, hipMemcpyHostToDevice, stream) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
            if (skip) {
                if ((rc = d_skip.alloc(cnt + 16))) break;
                if (cnt && hipMemcpyAsync(d_skip.p, skip + first, cnt, hipMemcpyHostToDevice, stream) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
            }
            if ((rc = katome_dist_add_reads(d, d_packed.as<uint8_t>(), first, cnt, read_len, skip ? d_skip.as<uint8_t>() : nullptr, 0, stream))) break;
            d_packed.release(); d_skip.release();
            // a rank whose reads could not be taken (memory, reads the route does not take) must not leave the others waiting inside
            // finalize's exchange: everybody meets here first, and a failed rank has poisoned the meeting
            if (!sync->barrier()) { set_error("another rank of this build failed"); rc = KATOME_E_DEVICE; break; }
            katome_dist_graph g;
            if ((rc = katome_dist_finalize(d, &g, stream))) break;
            if (first_seen && !direct) {
                katome_builder* root = nullptr;
                if ((rc = katome_dist_gather(d, 0, &root, stream))) break;
                if (r == 0) {
                    root->s.flags = s->flags; root->s.min_weight = s->min_weight;       // the stages the caller asked for run here
                    rc = finish(root, read_bytes);
                }
                break;
            }
            // the reference's numbering without a gather: remove_dead_paths (if asked for) on the sharded graph, then every rank
            // puts its edges and nodes at their indices in the host arrays
            bool pruned = false;
            if (first_seen && (s->flags & KATOME_FLAG_REMOVE_DEAD_PATHS)) {
                if ((rc = katome_dist_remove_dead_paths(d, &g, nullptr, stream))) break;
                pruned = true;
            }
            // by packed key: rank by rank
            sh.n_edges[r] = g.n_edges; sh.node_base[r] = g.node_base; sh.n_nodes[r] = g.n_nodes;
            if (!sync->barrier()) { set_error("another rank of this build failed"); rc = KATOME_E_DEVICE; break; }
            if (r == 0) {
                GraphOwner* o = new (std::nothrow) GraphOwner();
                if (!o) { set_error("out of host memory"); sh.alloc_rc = KATOME_E_OOM; }
                else {
                    memset(&o->g, 0, sizeof o->g);
                    katome_graph* hg = &o->g;
                    hg->n_nodes = g.total_nodes; hg->n_edges = g.total_edges; hg->read_bytes = read_bytes;
                    hg->k = s->k; hg->key_words = g.key_words; hg->label_stride = g.label_stride;
                    auto take = [&](size_t bytes) -> void* { void* q = host_result_alloc(bytes); if (q) o->mem.push_back(q); else sh.alloc_rc = KATOME_E_OOM; return q; };
                    hg->edge_src = (uint64_t*)take(std::max<uint64_t>(g.total_edges, 1) * 8);
                    hg->edge_dst = (uint64_t*)take(std::max<uint64_t>(g.total_edges, 1) * 8);
                    hg->edge_weight = (uint32_t*)take(std::max<uint64_t>(g.total_edges, 1) * 4);
                    hg->edge_label = (uint8_t*)take(std::max<uint64_t>(g.total_edges, 1) * (size_t)g.label_stride);
                    hg->edge_key = (uint64_t*)take(std::max<uint64_t>(g.total_edges, 1) * 8 * g.key_words);
                    hg->node_key = (uint64_t*)take(std::max<uint64_t>(g.total_nodes, 1) * 8 * g.key_words);
                    if (pruned) hg->edge_age = (uint32_t*)take(std::max<uint64_t>(g.total_edges, 1) * 4);
                    if (sh.alloc_rc) { set_error("out of host memory"); katome_graph_free(hg); o = nullptr; }
                }
                sh.owner = o;
            }
            if (!sync->barrier()) { set_error("another rank of this build failed"); rc = KATOME_E_DEVICE; break; }
            if (!sh.owner) { rc = sh.alloc_rc ? sh.alloc_rc : KATOME_E_OOM; break; }
            if (first_seen) {
                // every edge / node at its petgraph index: the rank's share comes over in its own order and is placed on the host
                katome_graph* hg = &sh.owner->g;
                const uint64_t E = g.n_edges, N = g.n_nodes;
                const uint32_t nwk = g.key_words, ls = g.label_stride;
                std::vector<uint64_t> id(E), src(E), dst(E), key(E * nwk), nid(N), nkey(N * nwk), age(pruned ? E : 0);
                std::vector<uint32_t> w(E);
                std::vector<uint8_t> lab(E * (size_t)ls);
                hipError_t e = hipSuccess;
                auto down = [&](void* h, const void* dv, size_t bytes) { if (bytes && e == hipSuccess) e = hipMemcpyAsync(h, dv, bytes, hipMemcpyDeviceToHost, stream); };
                down(id.data(), g.d_edge_id, E * 8); down(src.data(), g.d_edge_src, E * 8); down(dst.data(), g.d_edge_dst, E * 8);
                down(w.data(), g.d_edge_weight, E * 4); down(lab.data(), g.d_edge_label, E * (size_t)ls); down(key.data(), g.d_edge_key, E * 8 * nwk);
                down(nid.data(), g.d_node_id, N * 8); down(nkey.data(), g.d_node_key, N * 8 * nwk);
                if (pruned) down(age.data(), g.d_edge_age, E * 8);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (e != hipSuccess) { set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = KATOME_E_DEVICE; break; }
                uint64_t* h_src = const_cast<uint64_t*>(hg->edge_src); uint64_t* h_dst = const_cast<uint64_t*>(hg->edge_dst);
                uint32_t* h_w = const_cast<uint32_t*>(hg->edge_weight); uint8_t* h_lab = const_cast<uint8_t*>(hg->edge_label);
                uint64_t* h_key = const_cast<uint64_t*>(hg->edge_key); uint64_t* h_nkey = const_cast<uint64_t*>(hg->node_key);
                uint32_t* h_age = const_cast<uint32_t*>(hg->edge_age);
                bool bad = false;
                for (uint64_t i = 0; i < E; ++i) {
                    const uint64_t at = id[i];
                    if (at >= hg->n_edges) { bad = true; break; }
                    h_src[at] = src[i]; h_dst[at] = dst[i]; h_w[at] = w[i];
                    memcpy(h_lab + at * ls, lab.data() + i * (size_t)ls, ls);
                    for (uint32_t q = 0; q < nwk; ++q) h_key[at * nwk + q] = key[i * nwk + q];
                    if (pruned) { if (age[i] > 0xFFFFFFFFull) bad = true; h_age[at] = (uint32_t)age[i]; }
                }
                for (uint64_t j = 0; j < N && !bad; ++j) {
                    const uint64_t at = nid[j];
                    if (at >= hg->n_nodes) { bad = true; break; }
                    for (uint32_t q = 0; q < nwk; ++q) h_nkey[at * nwk + q] = nkey[j * nwk + q];
                }
                if (bad) { set_error("sharded build: an index does not fit the host result (katome_graph.edge_age is 32 bits wide)"); rc = KATOME_E_UNSUPPORTED; }
            } else {
                uint64_t e0 = 0;
                for (int p = 0; p < r; ++p) e0 += sh.n_edges[p];
                katome_graph* hg = &sh.owner->g;
                const uint64_t E = g.n_edges, N = g.n_nodes;
                const uint32_t nwk = g.key_words, ls = g.label_stride;
                hipError_t e = hipSuccess;
                if (E) {
                    if (e == hipSuccess) e = hipMemcpyAsync(const_cast<uint64_t*>(hg->edge_src) + e0, g.d_edge_src, E * 8, hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(const_cast<uint64_t*>(hg->edge_dst) + e0, g.d_edge_dst, E * 8, hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(const_cast<uint32_t*>(hg->edge_weight) + e0, g.d_edge_weight, E * 4, hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(const_cast<uint8_t*>(hg->edge_label) + e0 * ls, g.d_edge_label, E * (size_t)ls, hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(const_cast<uint64_t*>(hg->edge_key) + e0 * nwk, g.d_edge_key, E * 8 * nwk, hipMemcpyDeviceToHost, stream);
                }
                if (N && e == hipSuccess) e = hipMemcpyAsync(const_cast<uint64_t*>(hg->node_key) + g.node_base * nwk, g.d_node_key, N * 8 * nwk, hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (e != hipSuccess) { set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = KATOME_E_DEVICE; }
            }
        } while (0);
        if (d) katome_dist_destroy(d);
        dev_retire_stream(stream);
        (void)hipStreamDestroy(stream);
        return rc;
    };
    std::vector<std::thread> threads;
    for (int r = 0; r < n; ++r)
        threads.emplace_back([&, r]() {
            const int rc = body(r);
            sh.rc[r] = rc;
            if (rc) { sh.err[r] = get_error(); sync->poison(); if (group) group->poison(); }     // (ranks waiting at a host rendezvous give up)
        });
    for (auto& t : threads) t.join();
    for (int r = 0; r < n; ++r) katome_comm_destroy(comms[r]);
    int rc = KATOME_OK;
    for (int r = 0; r < n && !rc; ++r)
        if (sh.rc[r] && sh.err[r] != "another rank of this build failed") { rc = sh.rc[r]; set_error("rank %d of %d: %s", r, n, sh.err[r].c_str()); }
    for (int r = 0; r < n && !rc; ++r) if (sh.rc[r]) { rc = sh.rc[r]; set_error("rank %d of %d: %s", r, n, sh.err[r].c_str()); }
    if (!first_seen || direct) {
        if (rc == KATOME_OK && sh.owner && finish.graph) *finish.graph = &sh.owner->g;
        else if (sh.owner) katome_graph_free(&sh.owner->g);
    }
    return rc;
}

static int build_packed_impl(const katome_settings* s, const uint8_t* packed, uint64_t n_reads, uint32_t read_len,
                             const uint8_t* skip, const Finish& finish, const uint64_t* read_bytes_override = nullptr) {
    if (!s || (!packed && n_reads)) { set_error("null argument"); return KATOME_E_ARG; }
    KCHECK(check_k(s->k));
    // (KATOME_FORCE_SHARDED=1: one GPU through the sharded route as well -- a world of one rank with the transport a larger
    // world would use; testing)
    if ((s->n_devices > 1 || (s->n_devices == 1 && getenv("KATOME_FORCE_SHARDED"))) && n_reads && read_len >= s->k) {
        uint64_t read_bytes = 0;
        if (read_bytes_override) read_bytes = *read_bytes_override;
        else for (uint64_t r = 0; r < n_reads; ++r) if (!skip || !skip[r]) read_bytes += read_len;
        return build_packed_multi(s, packed, n_reads, read_len, skip, finish, read_bytes);
    }
    if (n_reads && read_len < s->k) {
        // only an ACCEPTED read can be too short (builder.rs:155-158 filters first)
        bool any = !skip;
        for (uint64_t r = 0; skip && r < n_reads && !any; ++r) any = skip[r] == 0;
        if (any) { set_error("Read is too short!"); return KATOME_E_SHORT_READ; }
        n_reads = 0;
    }
    katome_builder* b = nullptr;
    KCHECK(katome_builder_create(s, &b));
    build_lap("", true);
    int rc = KATOME_OK;
    do {
        const uint32_t stride = (read_len + 3) / 4, W = read_len >= s->k ? read_len - s->k + 1 : 0;
        uint64_t read_bytes = 0;
        for (uint64_t r = 0; r < n_reads; ++r) if (!skip || !skip[r]) read_bytes += read_len;
        DevBuf d_packed, d_skip, d_rec;
        if ((rc = d_packed.alloc(n_reads * stride + 32))) break;
        if (n_reads && hipMemcpy(d_packed.p, packed, n_reads * stride, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
        if (skip) {
            if ((rc = d_skip.alloc(n_reads + 16))) break;
            if (n_reads && hipMemcpy(d_skip.p, skip, n_reads, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
        }
        if (n_reads && W) {
            uint64_t reads_per_batch = std::max<uint64_t>(batch_records(b->nw) / W, 64);
            reads_per_batch = (reads_per_batch / 64) * 64;       // keeps batch starts 16-byte aligned
            if ((rc = d_rec.alloc(std::min(reads_per_batch, n_reads) * W * 8 * b->nw + 16))) break;
            uint32_t span = 1, tiles = 0, rest = 0;
            katome_tile_plan(s->k, read_len, &span, &tiles, &rest);
            for (uint64_t r0 = 0; r0 < n_reads && !rc; r0 += reads_per_batch) {
                const uint64_t nr = std::min(reads_per_batch, n_reads - r0);
                if (span > 1) {       // tiled counting: W/span tile records per read, then the windows that are left over
                    rc = katome_dev_count_tiles(b, d_packed.as<uint8_t>() + r0 * stride, nr, read_len, span,
                                                skip ? d_skip.as<uint8_t>() + r0 : nullptr, nullptr);
                    if (!rc && rest) {
                        rc = katome_dev_extract_remainder(b, d_packed.as<uint8_t>() + r0 * stride, nr, read_len, span,
                                                          skip ? d_skip.as<uint8_t>() + r0 : nullptr, d_rec.as<u64>(), nullptr);
                        if (!rc) rc = katome_dev_insert(b, d_rec.as<u64>(), nr * rest, nullptr);
                    }
                } else {
                    rc = katome_dev_extract_fixed(b, d_packed.as<uint8_t>() + r0 * stride, nr, read_len,
                                                  skip ? d_skip.as<uint8_t>() + r0 : nullptr, d_rec.as<u64>(), nullptr);
                    if (!rc) rc = katome_dev_insert(b, d_rec.as<u64>(), nr * W, nullptr);
                }
            }
            if (rc) break;
        }
        d_rec.release(); d_packed.release(); d_skip.release();
        rc = finish(b, read_bytes_override ? *read_bytes_override : read_bytes);
    } while (0);
    katome_builder_destroy(b);
    return rc;
}

extern "C" {

int katome_build_packed(const katome_settings* s, const uint8_t* packed, uint64_t n_reads, uint32_t read_len,
                        const uint8_t* skip, katome_graph** out) {
    if (!out) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    return build_packed_impl(s, packed, n_reads, read_len, skip, Finish{out, nullptr});
}
int katome_shrink_packed(const katome_settings* s, const uint8_t* packed, uint64_t n_reads, uint32_t read_len,
                         const uint8_t* skip, katome_contigs** out) {
    if (!out) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    return build_packed_impl(s, packed, n_reads, read_len, skip, Finish{nullptr, out});
}

int katome_ingest_files(const katome_settings* s, const char* const* paths, size_t n_paths, katome_reads** out) {
    if (!s || !out || (!paths && n_paths)) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    HostReads* hr = new (std::nothrow) HostReads();
    if (!hr) { set_error("out of host memory"); return KATOME_E_OOM; }
    int rc = ingest_files(s, paths, n_paths, *hr);
    if (rc) { delete hr; return rc; }
    struct Owner { katome_reads r; HostReads* hr; };
    Owner* o = new (std::nothrow) Owner();
    if (!o) { delete hr; set_error("out of host memory"); return KATOME_E_OOM; }
    o->hr = hr;
    o->r.n_records = hr->n_records; o->r.n_reads = hr->n_reads; o->r.read_bytes = hr->read_bytes;
    o->r.packed_bytes = hr->packed_bytes; o->r.total_windows = hr->total_windows; o->r.fixed_len = hr->fixed_len; o->r._pad = 0;
    o->r.packed = hr->packed; o->r.byte_off = hr->byte_off; o->r.len = hr->len;
    *out = &o->r;
    return KATOME_OK;
}
void katome_reads_free(katome_reads* r) {
    if (!r) return;
    struct Owner { katome_reads r; HostReads* hr; };
    Owner* o = reinterpret_cast<Owner*>(r);
    delete o->hr;
    delete o;
}

}  // extern "C"

static int build_files_impl(const katome_settings* s, const char* const* paths, size_t n_paths, const Finish& finish) {
    if (!s || (!paths && n_paths)) { set_error("null argument"); return KATOME_E_ARG; }
    HostReads hr;
    build_lap("", true);
    KCHECK(ingest_files(s, paths, n_paths, hr));           // path / parse / short-read errors surface before any GPU work
    build_lap("ingest (host)");
    if (s->file_type == 2) {
        // BFCounter (create_bfc, builder.rs:79-115; add_read_bfc, pt_graph.rs:317-330): every kept line is a k-mer
        // with a weight -> one edge per line and strand, exactly as add_single_edge_bfc (pt_graph.rs:201-213) adds them:
        // lines naming the same k-mer, and a k-mer that is its own reverse complement, stay parallel edges.
        katome_builder* b = nullptr;
        KCHECK(katome_builder_create(s, &b));
        int rc = KATOME_OK;
        do {
            if (hr.n_reads) {
                DevBuf d_packed, d_w, d_rec;
                if ((rc = d_packed.alloc(hr.packed_bytes + 32)) || (rc = d_w.alloc(hr.n_reads * 4)) || (rc = d_rec.alloc(hr.n_reads * 8 * b->nw + 16))) break;
                if (hipMemcpy(d_packed.p, hr.packed, hr.packed_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(d_w.p, hr.weight, hr.n_reads * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
                // the lines' k-mers as they are written (no canonical form: both strands become edges of their own)
                if ((rc = launch_extract_fixed(s->k, false, d_packed.as<uint8_t>(), hr.n_reads, s->k, nullptr, d_rec.as<u64>(), nullptr))) break;
                if ((rc = bfc_set_edges(b, d_rec.as<u64>(), d_w.as<u32>(), hr.n_reads, nullptr))) break;
                if (hipStreamSynchronize(nullptr) != hipSuccess) { set_error("device failure during build"); rc = KATOME_E_DEVICE; break; }
            }
            rc = finish(b, hr.read_bytes);
        } while (0);
        katome_builder_destroy(b);
        return rc;
    }
    if (hr.fixed_len) return build_packed_impl(s, hr.packed, hr.n_reads, hr.fixed_len, nullptr, finish, &hr.read_bytes);
    katome_builder* b = nullptr;
    KCHECK(katome_builder_create(s, &b));
    int rc = KATOME_OK;
    do {
        if (hr.n_reads == 0) { rc = finish(b, hr.read_bytes); break; }
        DevBuf d_packed, d_off, d_len, d_pref, d_rec;
        if ((rc = d_packed.alloc(hr.packed_bytes + 32)) || (rc = d_off.alloc((hr.n_reads + 1) * 8)) || (rc = d_len.alloc(hr.n_reads * 4))) break;
        if (hipMemcpy(d_packed.p, hr.packed, hr.packed_bytes, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_off.p, hr.byte_off, (hr.n_reads + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_len.p, hr.len, hr.n_reads * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
        uint64_t cap = batch_records(b->nw);
        if (const char* e = getenv("KATOME_VAR_BATCH_RECORDS")) cap = std::max<uint64_t>(1, strtoull(e, nullptr, 10));   // tests: many small batches
        // one span for the whole input: the one with the fewest table insertions over all reads (tiles from the front of
        // every read + the windows left over, as for fixed-length reads)
        uint32_t span = 1;
        if (!getenv("KATOME_NO_TILES")) {
            std::vector<uint64_t> hist;
            for (uint64_t r = 0; r < hr.n_reads; ++r) {
                const uint32_t W = hr.len[r] - s->k + 1;
                if (W >= hist.size()) hist.resize(W + 1, 0);
                hist[W] += 1;
            }
            uint64_t best = hr.total_windows;
            for (uint32_t sp = 2; sp <= 33 && s->k + sp - 1 <= 95; ++sp) {      // (the same charges as katome_tile_plan)
                bool breakable = sp <= 16;
                for (uint32_t d = 3; d <= 8 && !breakable; ++d) breakable = sp % d == 0;
                uint64_t cost = 0;
                for (size_t W = 1; W < hist.size(); ++W) cost += hist[W] * (W / sp + W % sp + (breakable ? 0 : 4));
                if (cost < best || (cost == best && span > 1)) { best = cost; span = sp; }
            }
            if (const char* e = getenv("KATOME_TILE_SPAN")) { const uint32_t sp = (uint32_t)atoi(e); if (sp >= 1 && s->k + sp - 1 <= 95) span = sp; }
        }
        DevBuf d_tpref, d_rpref;
        std::vector<uint64_t> pref, tpref, rpref;
        for (uint64_t r0 = 0; r0 < hr.n_reads && !rc;) {
            pref.assign(1, 0); tpref.assign(1, 0); rpref.assign(1, 0);
            uint64_t r1 = r0;
            while (r1 < hr.n_reads && (r1 == r0 || pref.back() + (hr.len[r1] - s->k + 1) <= cap)) {
                const uint64_t W = hr.len[r1] - s->k + 1;
                pref.push_back(pref.back() + W);
                tpref.push_back(tpref.back() + W / span);
                rpref.push_back(rpref.back() + W % span);
                ++r1;
            }
            const uint64_t windows = pref.back(), tiles = tpref.back(), rest = rpref.back(), nr = r1 - r0;
            if ((rc = d_pref.alloc(pref.size() * 8)) || (rc = d_rec.alloc(windows * 8 * b->nw + 16))) break;
            if (hipMemcpy(d_pref.p, pref.data(), pref.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
            if (span > 1) {
                if ((rc = d_tpref.alloc(tpref.size() * 8)) || (rc = d_rpref.alloc(rpref.size() * 8))) break;
                if (hipMemcpy(d_tpref.p, tpref.data(), tpref.size() * 8, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(d_rpref.p, rpref.data(), rpref.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { set_error("H2D copy failed"); rc = KATOME_E_DEVICE; break; }
                if (tiles) {
                    rc = katome_dev_extract_var_tiles(b, d_packed.as<uint8_t>(), hr.packed_bytes, d_off.as<u64>() + r0, d_len.as<u32>() + r0,
                                                      d_tpref.as<u64>(), d_pref.as<u64>(), nr, tiles, windows, span, d_rec.as<u64>(), nullptr);
                    if (!rc) rc = katome_dev_insert_tiles(b, d_rec.as<u64>(), tiles, span, nullptr);
                }
                if (!rc) rc = katome_dev_extract_var_remainder(b, d_packed.as<uint8_t>(), hr.packed_bytes, d_off.as<u64>() + r0, d_len.as<u32>() + r0,
                                                               d_rpref.as<u64>(), d_pref.as<u64>(), nr, rest, windows, span, d_rec.as<u64>(), nullptr);
                if (!rc) rc = katome_dev_insert(b, d_rec.as<u64>(), rest, nullptr);       // (also closes the batch when nothing is left over)
            } else {
                rc = katome_dev_extract_var(b, d_packed.as<uint8_t>(), hr.packed_bytes, d_off.as<u64>() + r0, d_len.as<u32>() + r0,
                                            d_pref.as<u64>(), nr, windows, d_rec.as<u64>(), nullptr);
                if (!rc) rc = katome_dev_insert(b, d_rec.as<u64>(), windows, nullptr);
            }
            if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) { set_error("device failure during build"); rc = KATOME_E_DEVICE; }
            r0 = r1;
        }
        if (rc) break;
        d_rec.release(); d_packed.release(); d_off.release(); d_len.release(); d_pref.release();
        rc = finish(b, hr.read_bytes);
    } while (0);
    katome_builder_destroy(b);
    return rc;
}

extern "C" {

int katome_build_files(const katome_settings* s, const char* const* paths, size_t n_paths, katome_graph** out) {
    if (!out) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    return build_files_impl(s, paths, n_paths, Finish{out, nullptr});
}
int katome_build_files_staged(const katome_settings* s, const char* const* paths, size_t n_paths, const char* stages,
                              uint64_t original_genome_length, katome_graph** out) {
    if (!out) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    Finish f{out, nullptr};
    f.stages = stages; f.genome_len = original_genome_length;
    return build_files_impl(s, paths, n_paths, f);
}
int katome_shrink_files(const katome_settings* s, const char* const* paths, size_t n_paths, katome_contigs** out) {
    if (!out) { set_error("null argument"); return KATOME_E_ARG; }
    *out = nullptr;
    return build_files_impl(s, paths, n_paths, Finish{nullptr, out});
}

// Stats<CollectionStats> for PtGraph (stats/collections.rs:137-168), from the host arrays
int katome_graph_stats(const katome_graph* g, katome_stats* st) {
    if (!g || !st) { set_error("null argument"); return KATOME_E_ARG; }
    memset(st, 0, sizeof *st);
    st->node_count = g->n_nodes; st->edge_count = g->n_edges;
    std::vector<uint32_t> outd(g->n_nodes, 0), ind(g->n_nodes, 0);
    uint64_t sum_w = 0;
    for (uint64_t e = 0; e < g->n_edges; ++e) {
        st->max_edge_weight = std::max(st->max_edge_weight, g->edge_weight[e]);
        sum_w += g->edge_weight[e];
        ++outd[g->edge_src[e]]; ++ind[g->edge_dst[e]];
    }
    st->avg_edge_weight = (double)sum_w / (double)g->n_edges;
    uint64_t sum_out = 0;
    for (uint64_t n = 0; n < g->n_nodes; ++n) {
        st->max_out_degree = std::max<uint64_t>(st->max_out_degree, outd[n]);
        st->max_in_degree = std::max<uint64_t>(st->max_in_degree, ind[n]);
        sum_out += outd[n];
        if (ind[n] == 0) ++st->incoming_vert_count;      // externals(Incoming)
        if (outd[n] == 0) ++st->outgoing_vert_count;     // externals(Outgoing)
    }
    st->avg_out_degree = (double)sum_out / (double)g->n_nodes;
    return KATOME_OK;
}

}  // extern "C"
