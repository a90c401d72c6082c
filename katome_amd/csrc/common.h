// common.h -- shared host-side plumbing of libkatome_gpu (error state, HIP checks, device buffers)
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#include "../../include/katome_gpu.h"
#include "kmer_bits.h"

namespace katome {

// ---- error state (katome_last_error) -------------------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

#define KCHECK_HIP(expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            katome::set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e),  \
                              __FILE__, __LINE__, #expr);                                        \
            return _e == hipErrorOutOfMemory ? KATOME_E_OOM : KATOME_E_DEVICE;                   \
        }                                                                                        \
    } while (0)

#define KCHECK(expr)                  \
    do {                              \
        int _rc = (expr);             \
        if (_rc != KATOME_OK) return _rc; \
    } while (0)

// caching allocator (mem.cpp): blocks are reused across phases and across builds
int dev_malloc(void** out, size_t bytes, hipStream_t stream);
void dev_free(void* p, hipStream_t stream);
size_t dev_cached_bytes();
// call before hipStreamDestroy: cached blocks remember the stream they were last used on
void dev_retire_stream(hipStream_t stream);
void dev_release_cache(int device);
void dev_cache_stats(int device, uint64_t out[3]);   // {bytes held from the driver, of them free, live blocks} on `device`

// gfx950 erratum (profiles/r03_shift64_erratum.md, tools/probe_shift64_top_vgpr.hip): v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64
// with the shift AMOUNT in the last VGPR of the wave's allocation sometimes shift by v0 instead (LLVM's Shift64HighRegBug, worked
// around by the compiler for gfx90a only).  The build checks every kernel's ISA for that shape (tools/scan_shift64_top_vgpr.py, run
// by the Makefile); a kernel it flags -- the amount in v<N>, N = 8n+7 -- puts KATOME_SHIFT64_GUARD(N+1) at its top: naming the next
// register makes the allocation a granule larger, so that the register after the amount exists.
#define KATOME_SHIFT64_GUARD(next_vgpr) asm volatile("" ::: "v" #next_vgpr)

// device buffer with RAII; `stream` is the stream the buffer's users are ordered on
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;
    DevBuf() {}
    explicit DevBuf(hipStream_t s) : stream(s) {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        int rc = dev_malloc(&p, n, stream);
        if (rc != KATOME_OK) { p = nullptr; return rc; }
        bytes = n;
        return KATOME_OK;
    }
    int alloc(size_t n, hipStream_t s) { stream = s; return alloc(n); }
    void release() {
        if (p) dev_free(p, stream);
        p = nullptr; bytes = 0;
    }
    void* take() { void* q = p; p = nullptr; bytes = 0; return q; }
    void adopt(void* q, size_t n) { release(); p = q; bytes = n; }
    template <class T> T* as() const { return (T*)p; }
};

// Optional HIP-event timing on the stream the kernels are launched on (bench.py's roofline figures).  PHASES bracket a step of
// the build (several launches); the K_* entries bracket ONE kernel launch each, so that a kernel's own average duration can be
// priced against its algorithmic bytes -- the free-standing primitives (radix.hip, table.hip) find the builder's profiler
// through a thread-local pointer that the enclosing PhaseScope sets.
enum Phase { PH_EXTRACT, PH_REGION_ORDER, PH_INSERT, PH_EMIT_EDGES, PH_SORT_EDGES, PH_NODE_SET, PH_RANK, PH_LABELS,
             PH_INSERT_TILES, PH_EXPAND_TILES, PH_EXPAND_MID, PH_FIRST_SEEN, PH_DEAD_PATHS, PH_SHRINK,
             K_SORT_SCATTER, K_SORT_HIST, K_RUN_SORT, K_HASH_SCATTER, K_HASH_HIST, K_OWNER_SCATTER, K_OWNER_HIST, K_PASS_OFFSETS,
             K_RECORDS, K_GROUP_INDEX, K_LDS_COUNT, K_SRC_IDS, K_DST_MERGE, K_EXPAND, K_SORT_SCATTER_KEYS, K_RUN_SORT_KEYS,
             K_TILE_HASH_SCATTER, K_TILE_HASH_HIST, K_TILE_RECORDS, K_TILE_GROUP_INDEX, K_TILE_LDS_COUNT, PH_COUNT };
static const char* const PHASE_NAMES[PH_COUNT] = {
    "extract", "region_order", "insert", "emit_edges", "sort_edges", "node_set", "rank", "labels", "insert_tiles", "expand_tiles",
    "expand_mid_tiles", "first_seen_order", "remove_dead_paths", "shrink",
    "k:radix_scatter_kernel<RadixDigit>", "k:radix_hist_kernel<RadixDigit>", "k:run_sort", "k:radix_scatter_kernel<HashDigit>",
    "k:radix_hist_kernel<HashDigit>", "k:radix_scatter_kernel<OwnerDigit>", "k:radix_hist_kernel<OwnerDigit>", "k:radix_chunk+offsets",
    "k:tiles_to_records_kernel", "k:hash_group_index_kernel", "k:lds_count_kernel", "k:src_count+src_write", "k:dst_merge_kernel",
    "k:expand_tiles_kernel",
    // (the keys-only instantiations -- the small sort of the nodes without out-edges -- are kernels of their own in a trace)
    "k:radix_scatter_kernel<RadixDigit> (keys only)", "k:run_sort (keys only)",
    // (the same kernels counting a TILE level by sorting -- other record sizes, so timed apart: TileLevelScope)
    "k:radix_scatter_kernel<HashDigit> (tile records)", "k:radix_hist_kernel<HashDigit> (tile records)", "k:tiles_to_records_kernel (tile records)",
    "k:hash_group_index_kernel (tile records)", "k:lds_count_kernel (tile records)"};
struct Profiler {
    bool on = false;
    struct Ev { int phase; hipEvent_t a, b; uint64_t work; };      // work: elements the launch processed (K_* entries)
    std::vector<Ev> evs;
    ~Profiler() { clear(); }
    void clear() { for (auto& e : evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); } evs.clear(); }
};
inline Profiler*& current_profiler() { static thread_local Profiler* p = nullptr; return p; }
struct PhaseScope {
    Profiler* p; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; int phase; Profiler* outer;
    PhaseScope(Profiler& prof, int ph, hipStream_t st) : p(prof.on ? &prof : nullptr), s(st), phase(ph), outer(current_profiler()) {
        if (p && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, s); else p = nullptr;
        if (p) current_profiler() = p;
    }
    ~PhaseScope() { if (p) { (void)hipEventRecord(b, s); p->evs.push_back({phase, a, b, 0}); current_profiler() = outer; } }
};
// while one of these is alive on the thread, the record/count kernels report to the K_TILE_* timers: a tile level counted by
// sorting runs the last level's kernels on records of another size, and a timer prices ONE record size
inline bool& counting_tile_level() { static thread_local bool on = false; return on; }
struct TileLevelScope {
    bool outer;
    TileLevelScope() : outer(counting_tile_level()) { counting_tile_level() = true; }
    ~TileLevelScope() { counting_tile_level() = outer; }
};
inline int tile_level_timer(int ph) {
    if (!counting_tile_level()) return ph;
    switch (ph) {
        case K_HASH_SCATTER: return K_TILE_HASH_SCATTER;
        case K_HASH_HIST:    return K_TILE_HASH_HIST;
        case K_RECORDS:      return K_TILE_RECORDS;
        case K_GROUP_INDEX:  return K_TILE_GROUP_INDEX;
        case K_LDS_COUNT:    return K_TILE_LDS_COUNT;
        default:             return ph;
    }
}
// one kernel launch (or a couple of tiny ones) inside a phase; a no-op unless a profiling PhaseScope is open on this thread
struct KernelScope {
    Profiler* p; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; int phase; uint64_t work;
    KernelScope(int ph, hipStream_t st, uint64_t elements = 0) : p(current_profiler()), s(st), phase(tile_level_timer(ph)), work(elements) {
        if (p && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, s); else p = nullptr;
    }
    ~KernelScope() { if (p) { (void)hipEventRecord(b, s); p->evs.push_back({phase, a, b, work}); } }
};

inline int use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): this library has no CPU fallback",
                  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        (void)hipGetLastError();
        return KATOME_E_DEVICE;
    }
    if (device < 0 || device >= n) { set_error("device ordinal %d out of range (0..%d)", device, n - 1); return KATOME_E_DEVICE; }
    KCHECK_HIP(hipSetDevice(device));
    return KATOME_OK;
}

inline int check_k(uint32_t k) {
    if (k <= 1) { set_error("assertion failed: k_size > 1"); return KATOME_E_ARG; }   // prelude.rs:35
    if (k < 3 || k > 63) { set_error("k = %u unsupported (3..63)", k); return KATOME_E_UNSUPPORTED; }   // valid for the reference, not here
    return KATOME_OK;
}

constexpr int BLOCK = 256;
inline unsigned grid_for(uint64_t work_items, unsigned per_block, unsigned cap = 256u * 16u) {
    uint64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

// ---- launchers implemented in the .hip files (all asynchronous on `stream`) -----------------
int launch_extract_fixed(uint32_t k, bool rc, const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len,
                         const uint8_t* d_skip, uint64_t* d_records, hipStream_t stream, uint32_t span = 1, bool mark = false,
                         uint32_t first_window = 0, uint32_t records_per_read = 0);
int launch_extract_var(uint32_t k, bool rc, const uint8_t* d_packed, uint64_t packed_bytes, const uint64_t* d_byte_off,
                       const uint32_t* d_len, const uint64_t* d_win_prefix, uint64_t n_reads, uint64_t total_windows,
                       uint64_t* d_records, hipStream_t stream, bool mark = false, uint32_t span = 1, uint32_t mode = 0);

// radix.hip
int dev_sort(uint64_t* d_keys, uint32_t* d_vals, uint64_t n, uint32_t nw, uint32_t key_bits, hipStream_t stream);
struct DevBuf;
// (distinct_keys: no two keys are equal -- the first pass then need not be stable: radix.hip)
int dev_sort_bufs(DevBuf& keys, DevBuf* vals, uint64_t n, uint32_t nw, uint32_t key_bits, hipStream_t stream, bool distinct_keys = false);
int dev_partition(const uint64_t* d_in, const uint32_t* v_in, uint64_t n, uint32_t nw, uint32_t n_parts, uint64_t* d_out,
                  uint32_t* v_out, uint64_t* h_counts, hipStream_t stream, uint32_t core_shift = 0, uint32_t core_bases = 0, uint32_t minimizer = 0);
// supermer.hip: the sharded build's exchange unit
constexpr uint32_t SUPERMER_M = 11;        // bases of the minimizer that names a k-mer's owner on the supermer route
uint32_t supermer_slots(uint32_t k, uint32_t read_len, uint32_t m);
bool supermer_route_takes(uint32_t k, uint32_t read_len, uint32_t m);
int dev_supermers_extract(const uint8_t* d_packed, uint64_t n_reads, uint32_t read_len, const uint8_t* d_skip, uint32_t k, uint32_t m, bool rc,
                          uint32_t n_owners, uint32_t slots, uint64_t* d_out, uint64_t* d_spill, uint64_t spill_cap, uint64_t* d_spill_cursor,
                          hipStream_t stream);
int dev_supermers_expand(const uint64_t* d_list, const uint32_t* d_counts, uint64_t n, uint32_t k, bool rc, DevBuf& keys, DevBuf& weights,
                         uint64_t* n_records, hipStream_t stream);
int dev_partition_supermers(const uint64_t* d_in, uint64_t n, uint32_t n_parts, uint64_t* d_out, uint64_t* h_counts, hipStream_t stream);
int dev_partition_range(const uint64_t* d_vals, const uint32_t* idx_in, uint64_t n, const uint64_t* d_bounds, uint32_t n_parts,
                        uint64_t* d_out, uint32_t* idx_out, uint64_t* h_counts, hipStream_t stream);
int dev_source_ids(const uint64_t* d_edge_key, uint64_t n_edges, uint32_t k, DevBuf& node_key, uint64_t* d_edge_src, uint64_t* n_src,
                   hipStream_t stream);
int dev_hash_order_core(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t core_shift, uint32_t core_bases, uint64_t* ka, uint64_t* kb,
                        uint32_t* wa, uint32_t* wb, const uint64_t** k_out, const uint32_t** w_out, uint32_t* group_bits, hipStream_t stream);
int dev_hash_order_tagged(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t nwk, uint64_t* ka, uint64_t* kb, uint32_t* wa, uint32_t* wb,
                          const uint64_t** k_out, const uint32_t** w_out, uint32_t* group_bits, hipStream_t stream, const uint32_t* first_counts = nullptr);
int dev_region_order(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t nw, int passes, uint64_t* ka, uint64_t* kb,
                     uint32_t* wa, uint32_t* wb, const uint64_t** k_out, const uint32_t** w_out, hipStream_t stream);
int dev_hash_order(const uint64_t* d_in, const uint32_t* w_in, uint64_t n, uint32_t nw, uint64_t* ka, uint64_t* kb, uint32_t* wa, uint32_t* wb,
                   const uint64_t** k_out, const uint32_t** w_out, uint32_t* group_bits, hipStream_t stream, const uint32_t* first_counts = nullptr);
uint32_t dev_sort_tile_keys(uint32_t nw);
int dev_unique(uint64_t* d_keys, uint64_t n, uint32_t nw, uint64_t* n_out, hipStream_t stream);
int dev_rank(const uint64_t* d_sorted, uint64_t n_sorted, uint32_t nw, uint32_t key_bits, const uint64_t* d_q, uint64_t nq,
             uint64_t* d_out, hipStream_t stream);
// d_seq + node_first (first-seen order): also the nodes' first touches (dev_node_first's result), filled on the way; node_first
// comes back EMPTY when that was not done and dev_node_first has to run
int dev_node_ids(const uint64_t* d_edge_key, uint64_t n_edges, uint32_t k, DevBuf& node_key, uint64_t* d_edge_src,
                 uint64_t* d_edge_dst, uint64_t* n_nodes, hipStream_t stream, const uint64_t* d_seq = nullptr, DevBuf* node_first = nullptr,
                 uint64_t* n_marked = nullptr);
int dev_iota(uint32_t* d, uint64_t n, hipStream_t stream);
int dev_fill_u32(uint32_t* d, uint64_t n, uint32_t v, hipStream_t stream);
int dev_gather_seq_weight(const uint64_t* pairs, const uint32_t* idx, uint64_t n, uint64_t* seq, uint32_t* weight, hipStream_t stream);
int dev_gather_u32(const uint32_t* src, const uint32_t* idx, uint64_t n, uint32_t* dst, hipStream_t stream);
int dev_gather_u64(const uint64_t* src, const uint32_t* idx, uint64_t n, uint64_t* dst, hipStream_t stream);
int dev_gather_keys(const uint64_t* src, const uint32_t* idx, uint64_t n, uint32_t nw, uint64_t* dst, hipStream_t stream);
int dev_gather_mapped(const uint64_t* src, const uint32_t* idx, const uint64_t* map, uint64_t n, uint64_t* dst, hipStream_t stream);
int dev_invert(const uint32_t* perm, uint64_t n, uint64_t* inv, hipStream_t stream);
int dev_permute_edges(uint64_t* key, uint32_t* weight, uint64_t* src, uint64_t* dst, const uint64_t* new_id, const uint32_t* idx,
                      uint64_t n, uint32_t nw, void* scratch, hipStream_t stream);
int dev_pack_edges_intro(const uint64_t* key, const uint32_t* weight, const uint64_t* src, const uint64_t* dst, const uint64_t* seq,
                         const uint64_t* node_first, uint64_t n, uint32_t nw, void* aos, hipStream_t stream, uint64_t n_marked = 0);
int dev_unpack_edges_intro(const void* aos, const uint32_t* idx, uint64_t n, uint32_t nw, uint64_t* key, uint32_t* weight, uint64_t* src,
                           uint64_t* dst, uint32_t* cnt, hipStream_t stream);
int dev_assign_nodes(const uint64_t* key, const uint64_t* src, const uint64_t* dst, const uint64_t* offs, uint64_t n, uint32_t nw, uint32_t k,
                     uint64_t* new_id, uint64_t* node_key, uint64_t* out_src, uint64_t* out_dst, hipStream_t stream);
int dev_clear_dst_marks(uint64_t* dst, uint64_t n, hipStream_t stream);
int dev_node_first(const uint64_t* src, const uint64_t* dst, const uint64_t* seq, uint64_t n, uint64_t* node_first, hipStream_t stream);
int dev_endpoints(const uint64_t* d_edge_key, uint64_t n, uint32_t k, uint64_t* d_src, uint64_t* d_dst, hipStream_t stream);
int dev_labels(const uint64_t* d_edge_key, uint64_t n, uint32_t k, uint8_t* d_label, hipStream_t stream);
// BFCounter lines -> one edge per line (and per strand), unmerged (pt_graph.rs:201-213); seq (nullable) = petgraph index
int dev_bfc_edges(const uint64_t* d_fwd, const uint32_t* d_w, uint64_t n, uint32_t k, bool rc, uint64_t* d_edge_key,
                  uint32_t* d_edge_weight, uint64_t* d_edge_seq, hipStream_t stream);
// exclusive scan of m u32 counts into u64 offsets (offs[m] = total), one workgroup
int dev_scan_counts(const uint32_t* d_counts, uint64_t m, uint64_t* d_offs, hipStream_t stream);

// prune.hip: Prunable::remove_dead_paths on a first-seen-ordered graph, in place (counts shrink, buffers stay)
struct PruneGraph {
    DevBuf *edge_src, *edge_dst, *edge_weight, *edge_key, *node_key;
    DevBuf *edge_age;      // u32 per edge: its first-seen index = its place in petgraph's adjacency lists (swap_remove re-labels
                           // edges but never reorders the lists); empty = the edges still sit at their first-seen positions
    uint64_t n_edges, n_nodes;
    uint32_t nw;
    bool parallel_edges = false;   // BFCounter graphs may hold the same k-mer on several edges: no per-base adjacency slots then
};
// scratch of dev_replay_edges, kept across the passes of remove_dead_paths
struct ReplayScratch {
    DevBuf agg, carry, size_before, jump, totals;
    explicit ReplayScratch(hipStream_t stream) : agg(stream), carry(stream), size_before(stream), jump(stream), totals(stream) {}
};
int dev_replay_edges(const uint32_t* d_pos, const uint32_t* d_mult, uint64_t u, uint64_t n_edges, ReplayScratch& scratch, DevBuf& victims,
                     DevBuf& move_to, DevBuf& move_from, uint64_t* n_removed, uint64_t* n_left, uint64_t* n_moves, uint64_t* n_dups,
                     hipStream_t stream);
int dev_replay_edges64(const uint64_t* d_pos, const uint32_t* d_mult, uint64_t u, uint64_t n_edges, ReplayScratch& scratch, DevBuf& victims,
                       DevBuf& move_to, DevBuf& move_from, uint64_t* n_removed, uint64_t* n_left, uint64_t* n_moves, uint64_t* n_dups,
                       hipStream_t stream);
struct NodeReplayScratch {
    DevBuf counts, offs, first, hole, dead, flags;
    explicit NodeReplayScratch(hipStream_t stream) : counts(stream), offs(stream), first(stream), hole(stream), dead(stream), flags(stream) {}
};
int dev_replay_nodes(const uint32_t* d_die, uint64_t m, uint64_t n_nodes, NodeReplayScratch& scratch, DevBuf& move_to, DevBuf& move_from,
                     uint64_t* n_moves, uint64_t* n_left, int* fell_back, hipStream_t stream);
int dev_replay_nodes64(const uint64_t* d_die, uint64_t m, uint64_t n_nodes, NodeReplayScratch& scratch, DevBuf& move_to, DevBuf& move_from,
                       uint64_t* n_moves, uint64_t* n_left, int* fell_back, hipStream_t stream);
int dev_remove_dead_paths(PruneGraph& g, uint32_t k, katome_prune_stats* st, hipStream_t stream);
// Clean::remove_weak_edges with petgraph's retain_edges / retain_nodes numbering, same graph, in place
int dev_remove_weak_edges_ordered(PruneGraph& g, uint32_t threshold, hipStream_t stream);

// standardize.hip: Standardizable (standardizer.rs:41-128)
int dev_standardize_contigs(const uint64_t* src, const uint64_t* dst, uint32_t* weight, uint64_t E, uint64_t N, hipStream_t stream);
int dev_standardize_scale(uint32_t* weight, uint64_t E, uint64_t original_genome_length, uint32_t k, uint32_t threshold, hipStream_t stream);

// shrink.hip: Shrinkable::shrink (shrinker.rs:165-209) on a finalized graph; the result lives in its own buffers
struct ShrinkInput {
    const uint64_t *edge_src, *edge_dst; const uint32_t* edge_weight; const uint64_t *edge_key, *node_key;
    uint64_t n_edges, n_nodes; uint32_t nw, k;
};
struct ShrinkOutput {
    DevBuf edge_src, edge_dst, edge_weight, edge_kmers, edge_label_off, edge_label, node_key;
    uint64_t n_edges = 0, n_nodes = 0, label_bytes = 0;
};
int dev_shrink(const ShrinkInput& g, ShrinkOutput& out, hipStream_t stream);
// the reference's own result: its cuts and petgraph's numbering (shrink_exact.h on the host for the order, the device for the bytes)
int dev_shrink_exact(const ShrinkInput& g, const uint32_t* edge_age, ShrinkOutput& out, double* host_ms, hipStream_t stream);

constexpr int KATOME_MAX_RANKS = 16;      // ranks of a sharded build (an MI355X node has 8 GPUs)

// table.hip
struct Table {
    DevBuf slots;          // NW=1: {u64 key|OCC, u32 count, u32 pad}; NW=2: {u64 hi|flags, u64 lo, u32 count, u32 pad[3]}
    DevBuf counter;        // u64 occupied
    DevBuf seen;           // first-seen-order mode: [cap][2] u64 earliest sequence base per strand (table.hip)
    bool track_seen = false;
    uint64_t cap = 0;
    uint32_t nw = 1;
    size_t slot_bytes() const { return nw == 1 ? 16 : 32; }
    void release() { slots.release(); counter.release(); seen.release(); cap = 0; }      // (cap == 0: "no table" -- expand_to_last_level's test)
};
// where a batch of records sits in the read-ordered stream (first-seen-order mode)
struct SeenOrigin {
    uint64_t read0 = 0, rec0 = 0; uint32_t per_read = 1, span = 1, windows = 1; bool rc = false;
    uint32_t win0 = 0;                 // first window the batch's records cover in every read (the windows after the tiles)
    // variable-length reads: record g of the batch is window g - win_prefix[r] of the read r with win_prefix[r] <= g;
    // seq_base = sequence numbers used by the batches before this one
    const uint64_t* win_prefix = nullptr; uint64_t n_reads = 0, seq_base = 0;
    // ... whose records may be whole tiles (mode 1) or the windows after them (mode 2): rec_prefix = records before each read
    const uint64_t* rec_prefix = nullptr; uint32_t mode = 0;
    // sharded build: the records arrive grouped by the rank that extracted them (n_seg segments, segment p = records
    // [seg_off[p], seg_off[p+1]) of the batch, cut from reads seg_read0[p]...), each with its index in THAT rank's batch ...
    const uint32_t* idx = nullptr; uint32_t n_seg = 0;
    uint64_t seg_off[KATOME_MAX_RANKS + 1] = {0}, seg_read0[KATOME_MAX_RANKS] = {0};
    // ... or with both sequence numbers spelled out ([n][2]: stored orientation, its reverse complement): k-mer records made
    // out of another rank's tiles
    const uint64_t* pairs = nullptr;
};
int table_alloc(Table& t, uint32_t nw, uint64_t cap, hipStream_t stream);
int table_insert(Table& t, const uint64_t* d_records, const uint32_t* d_weights, uint64_t n, hipStream_t stream,
                 const SeenOrigin* origin = nullptr);
int table_occupied(Table& t, uint64_t* out, hipStream_t stream);
int table_grow(Table& t, uint64_t new_cap, hipStream_t stream);
// tiled counting: every tile (key of k+span-1 bases, weight n) adds n to each of its `span` k-mers
int table_expand_tiles(Table& tiles, uint64_t slot0, uint64_t slot1, Table& kmers, uint32_t k, uint32_t span, uint32_t stride,
                       bool rc, hipStream_t stream);
// same, but the (k-mer, weight) records are written out instead (multi-GPU: they travel to their owners)
// (seen: when the tile table tracks first-seen order, the records' two sequence numbers, [n][2])
int table_expand_tiles_to_records(Table& tiles, uint32_t k, uint32_t span, bool rc, DevBuf& keys, DevBuf& weights,
                                  uint64_t* n_records, hipStream_t stream, DevBuf* seen = nullptr);
int table_list_to_records(const uint64_t* d_tiles, const uint32_t* d_counts, uint64_t n_tiles, uint32_t tile_bases, uint32_t k, uint32_t span, uint32_t stride, bool rc,
                          DevBuf& keys, DevBuf& weights, uint64_t* n_records, hipStream_t stream, uint64_t extra_room = 0, DevBuf* first_counts = nullptr);
// (first_counts: filled with the digit counts per tile of the first partition pass over these records -- dev_hash_order's first_counts --
// when the kernel can make them as it writes; released otherwise)
// (extra_room: records the caller will append behind them -- the windows left over after the tiles)
int table_tiles_to_records_fast(Table& tiles, uint32_t k, uint32_t span, bool rc, DevBuf& keys, DevBuf& weights, uint64_t* n_records, hipStream_t stream,
                                uint64_t extra_room = 0);
// first-seen builds, last level counted by sorting: the tiles' k-mers as (k-mer, packed sequence numbers) records, counted and
// numbered in LDS; edge_key (unsorted) and seq_weight ([n][2]: {sequence number, weight}) come out as table_emit_edges leaves them.
// seq_per_read: sequence numbers a read takes (2 x windows); E_UNSUPPORTED when the packing does not fit (the caller counts in the table)
// (d_extra / n_extra: tagged records to count with them -- the windows left over after the tiles, table_rest_to_tagged)
int tiles_to_edges_sorted_seen(Table& tiles, uint32_t k, uint32_t span, bool rc, uint64_t seq_per_read, DevBuf& edge_key, DevBuf& seq_weight,
                               uint64_t* n_edges, uint64_t* n_distinct, hipStream_t stream, const uint64_t* d_extra = nullptr, uint64_t n_extra = 0);
// appends the valid records of d_rec behind *d_cursor in d_out (tagged: as records of nw + 1 words, see seen_pack in table.hip)
int table_keep_rest(const uint64_t* d_rec, uint64_t n, uint32_t nw, bool tagged, uint64_t read0, uint32_t per_read, uint32_t win0, uint32_t seq_per_read,
                    uint64_t* d_out, uint64_t* d_cursor, hipStream_t stream, uint32_t win_stride = 1, uint32_t span = 1);
int tagged_records_sorted(DevBuf& recs, DevBuf& wts, uint64_t n, uint32_t k, bool rc, uint64_t seq_per_read, bool list, DevBuf& out_keys,
                          DevBuf& out_second, uint64_t* n_out, uint64_t* n_distinct, hipStream_t stream, const uint32_t* first_counts = nullptr);
int table_list_to_tagged_records(const uint64_t* d_list, const uint32_t* d_counts, uint64_t n_tiles, uint32_t tile_bases, uint32_t sub_len, uint32_t n_sub,
                                 uint32_t stride, bool rc, DevBuf& recs, DevBuf& weights, uint64_t* n_records, hipStream_t stream, uint64_t extra_room = 0,
                                 DevBuf* first_counts = nullptr);
int table_tagged_to_pairs(const uint64_t* d_tagged, uint64_t n, uint32_t nw, uint64_t seq_per_read, uint64_t* d_keys, uint64_t* d_pairs, hipStream_t stream);
// A rank's own distinct k-mers on their way to their owners (sharded build): with an OwnerSplit the records are grouped by the hash
// of their CORE instead of the whole k-mer's, and every group's keys are written into its owner's stretch of the output -- base[p],
// count[p] on return --, so that no partition pass is needed before the exchange.  One-word k-mers, one record per k-mer (rc = false).
struct OwnerSplit {
    uint32_t n_parts = 0, core_shift = 0, core_bases = 0;
    uint64_t base[KATOME_MAX_RANKS] = {0}, count[KATOME_MAX_RANKS] = {0};
};
int records_to_edges_sorted(DevBuf& keys, DevBuf& weights, uint64_t n, uint32_t k, bool rc, uint32_t min_weight, DevBuf& edge_key,
                            DevBuf& edge_weight, uint64_t* n_edges, uint64_t* n_distinct, hipStream_t stream, OwnerSplit* split = nullptr,
                            const uint32_t* first_counts = nullptr);
int table_to_records(Table& t, DevBuf& keys, DevBuf& weights, uint64_t* n_records, hipStream_t stream, DevBuf* seen_pairs = nullptr);
int table_expand_tiles_to_subtiles(Table& tiles, uint32_t k, uint32_t span, uint32_t stride, bool rc, DevBuf& keys, DevBuf& weights,
                                   uint64_t* n_records, hipStream_t stream, DevBuf* seen = nullptr);
// distinct oriented edges (unsorted): allocates d_keys/d_weights
int table_emit_edges(Table& t, uint32_t k, bool rc, uint32_t min_weight, DevBuf& keys, DevBuf& weights, uint64_t* n_edges,
                     hipStream_t stream, DevBuf* seqs = nullptr);

// synth.hip
int launch_synth(uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len, double err_rate,
                 uint32_t n_inject_percent, uint8_t* d_packed, uint8_t* d_skip, hipStream_t stream);

// ingest.cpp (host only)
struct HostReads {
    uint64_t n_records = 0, n_reads = 0, read_bytes = 0, total_windows = 0;
    uint32_t fixed_len = 0;
    bool all_fixed = true;
    uint8_t* packed = nullptr; uint64_t packed_bytes = 0, packed_cap = 0;
    uint64_t* byte_off = nullptr; uint32_t* len = nullptr; uint64_t cap_reads = 0;
    uint32_t* weight = nullptr;          // BFCounter input only: one weight per k-mer line (builder.rs:96-105)
    ~HostReads();
};
int ingest_files(const katome_settings* s, const char* const* paths, size_t n_paths, HostReads& out);

}  // namespace katome
