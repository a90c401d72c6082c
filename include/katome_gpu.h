/*
 * katome_gpu.h -- C ABI of the MI355X-native replacement for katome's `build` stage.
 *
 * Drop-in boundary: `Build::create(input_files, ft, reverse_complement, minimal_weight_threshold)
 * -> (Self, usize)` (reference src/katome/algorithms/builder.rs:42-54), called by
 * `BasicAsm::assemble` / `assemble_with_gir` (asm/basic_assembler.rs:22-26, 37-40) after
 * `set_global_k_sizes(config.k_mer_size)` (basic_assembler.rs:19-21, prelude.rs:34-43).
 * A Rust `GpuGIR` type binds these symbols with `extern "C"` and implements
 * `Convert<GpuGIR> for PtGraph` (collections/girs/mod.rs:16-29) from the arrays of `katome_graph`.
 *
 * Plain pointers and sizes only; nothing unwinds across the boundary (the reference panics:
 * builder.rs:62,67,71,124,148,153, pt_graph.rs:278 -- here every entry returns a status code and
 * `katome_last_error()` holds the message the reference would have panicked with).
 *
 * Threading: synchronous, one build at a time per process (the reference keeps k in `static mut`
 * globals, prelude.rs:21-25); the library uses HIP streams internally.
 */
#ifndef KATOME_GPU_H
#define KATOME_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KATOME_ABI_VERSION 2

/* settings.flags.  FIRST_SEEN_ORDER: number edges and nodes exactly as the reference's sequential loop does --
 * petgraph edge indices in the order `add_edge` is first called for them, node indices in the order `add_node`
 * is (pt_graph.rs:149,194) -- instead of by packed key.  Downstream stages that walk petgraph's adjacency
 * (pruner.rs:241, collapser.rs:120) then see the same graph, index for index.  Costs two extra atomics per
 * counted record and two more sorts at the end; one GPU (reads of unequal length go through the general
 * one-record-per-window route).                                                                     */
#define KATOME_FLAG_FIRST_SEEN_ORDER 1u
/* REMOVE_DEAD_PATHS (host entries katome_build_*): run Prunable::remove_dead_paths (pruner.rs:36-82) on the built
 * graph before it is handed back, as assemble() does right after the build (asm/basic_assembler.rs:58-62).  The reference's
 * walks and swap-removes depend on petgraph's numbering, so this needs FIRST_SEEN_ORDER as well (KATOME_E_ARG
 * otherwise); the result then equals the reference's pruned PtGraph index for index.                      */
#define KATOME_FLAG_REMOVE_DEAD_PATHS 2u
/* n_devices > 1 only: all ranks of the sharded build run on settings.device (peer copies instead of RCCL, which refuses two
 * ranks on one GPU).  Exercises the whole multi-GPU route -- sharding, routing, exchanges, global numbering -- on a
 * one-GPU box; no performance meaning.                                                                       */
#define KATOME_FLAG_RANKS_SHARE_DEVICE 4u

/* status codes: the reference's panics, one code each */
enum {
    KATOME_OK = 0,
    KATOME_E_PATH = -1,        /* builder.rs:62  "Coulndt resolve path: ..."            */
    KATOME_E_IS_DIR = -2,      /* builder.rs:67  "... is a directory"                   */
    KATOME_E_NOT_EXIST = -3,   /* builder.rs:71  "... does not exist"                   */
    KATOME_E_OPEN = -4,        /* builder.rs:124,148 "Couldn't open all files: ..."     */
    KATOME_E_PARSE = -5,       /* builder.rs:128,153 record `.unwrap()` on a bad record */
    KATOME_E_SHORT_READ = -6,  /* pt_graph.rs:278 / hm_gir.rs:40 "Read is too short!"   */
    KATOME_E_ARG = -7,         /* prelude.rs:35 `assert!(k_size > 1)`, bad settings     */
    KATOME_E_DEVICE = -8,      /* no usable MI355X / HIP error: there is NO CPU fallback */
    KATOME_E_OOM = -9,
    KATOME_E_UNSUPPORTED = -10
};

/* Config<P> (config.rs:18-37) + prelude globals, as a plain struct */
typedef struct {
    uint32_t k;                  /* k_mer_size; supported 3..63 (k<=31: 64-bit keys, else 128-bit) */
    uint8_t  file_type;          /* InputFileType (config.rs:5-14): 0 Fasta, 1 Fastq, 2 BFCounter   */
    uint8_t  reverse_complement; /* Config::reverse_complement                                     */
    uint16_t flags;              /* KATOME_FLAG_*                                                   */
    uint32_t min_weight;         /* minimal_weight_threshold (BFCounter ingest only, builder.rs:106)*/
    int32_t  device;             /* HIP device ordinal of the (first) MI355X to build on             */
    uint64_t table_slots_hint;   /* 0 = size the (k-1)-mer/edge table automatically (whole build)   */
    /* GPUs of this node to build on: 0 or 1 = settings.device alone; n > 1 = devices device .. device+n-1,
     * one host thread and one rank per GPU INSIDE this call, reads sharded by index, records routed to their owners over
     * RCCL (katome_build_files / katome_build_packed; fixed-length reads -- other inputs are built on `device` alone).
     * The result does not depend on n in FIRST_SEEN_ORDER (the reference's numbering); by packed key, edges and nodes
     * come rank by rank (every rank's in ascending order).                                                   */
    int32_t  n_devices;
    uint32_t _reserved;
} katome_settings;

/* Result of a build, host memory, owned by the library until katome_graph_free().
 * What `Convert<GpuGIR> for PtGraph` needs (pt_graph.rs:33): node count, and per edge
 * (source, target, weight, label) with the label in compress_edge format (compress.rs:250-271),
 * i.e. what SEQUENCES[EdgeSlice.idx()] holds after PtGraph::create (pt_graph.rs:339-343).
 * Edge order = ascending packed k-mer.  Node ids (deterministic): the nodes that have out-edges in
 * ascending packed-(k-1)-mer order, then the nodes without out-edges in ascending order.
 * With KATOME_FLAG_FIRST_SEEN_ORDER: edges and nodes in the reference's own (first-seen) order.    */
typedef struct {
    uint64_t n_nodes, n_edges;
    uint64_t read_bytes;          /* sum of accepted read lengths (builder.rs:158)              */
    uint32_t k;
    uint32_t key_words;           /* u64 words per key: 1 (k<=31) or 2                          */
    uint32_t label_stride;        /* 1 + ceil(k/4)                                              */
    uint32_t _pad;
    const uint64_t *edge_src;     /* [n_edges] dense node id of the source (k-1)-mer            */
    const uint64_t *edge_dst;     /* [n_edges] dense node id of the target (k-1)-mer            */
    const uint32_t *edge_weight;  /* [n_edges] EdgeWeight = u32 (prelude.rs:9), wrapping        */
    const uint8_t  *edge_label;   /* [n_edges][label_stride] compress_edge format               */
    const uint64_t *edge_key;     /* [n_edges][key_words] packed k-mer, right-aligned, w[0] = high word */
    const uint64_t *node_key;     /* [n_nodes][key_words] packed (k-1)-mer, right-aligned       */
    /* FIRST_SEEN_ORDER graphs after remove_dead_paths / remove_weak_edges (NULL otherwise: age == index): the
     * first-seen index every edge had BEFORE the removals re-numbered it.  petgraph's swap_remove re-labels edges but
     * keeps every adjacency list in insertion order, so the reference's graph at this point is "these indices, lists
     * ordered by age" (first_edge = the live edge with the largest age); see INTEGRATION.md                    */
    const uint32_t *edge_age;
} katome_graph;

/* CollectionStats (stats/collections.rs:38-57) as computed for PtGraph (137-168) */
typedef struct {
    uint64_t node_count, edge_count;
    uint32_t max_edge_weight;
    uint32_t _pad;
    double   avg_edge_weight;
    uint64_t max_in_degree, max_out_degree;
    double   avg_out_degree;
    uint64_t incoming_vert_count, outgoing_vert_count;
} katome_stats;

/* ---- the drop-in entry points --------------------------------------------------------- */

/* Build::create over files (builder.rs:42-54 -> check_files 57-77 -> create_fastq 142-165 /
 * create_fasta 118-140): parse, skip reads with a non-ACGT byte, pack 2 bits/base, build on
 * the GPU, copy the graph back. */
int katome_build_files(const katome_settings *s, const char *const *paths, size_t n_paths,
                       katome_graph **out);
/* the build followed by stages of assemble_with_graph (asm/basic_assembler.rs:58-72), on the device, in the order
 * `stages` names them: 'd' Prunable::remove_dead_paths, 'c' Standardizable::standardize_contigs, 'w'
 * Clean::remove_weak_edges(settings.min_weight), 'e' standardize_edges(original_genome_length, k, settings.min_weight).
 * "dcwced" is everything the reference does before collapse().  Needs KATOME_FLAG_FIRST_SEEN_ORDER; the result carries
 * edge_age (see katome_graph).                                                                              */
int katome_build_files_staged(const katome_settings *s, const char *const *paths, size_t n_paths, const char *stages,
                              uint64_t original_genome_length, katome_graph **out);

/* Same build from reads that are already 2-bit packed (A0 C1 G2 T3, 4 bases per byte, first
 * base in the two most significant bits: the bit order of compress_node, compress.rs:55-73).
 * Read r occupies bytes [r*ceil(read_len/4), +ceil(read_len/4)).  `skip` (nullable) has one
 * byte per read; non-zero = the read held a non-ACGT byte and is not added (builder.rs:155-157).
 * read_len < k is KATOME_E_SHORT_READ (pt_graph.rs:278). */
int katome_build_packed(const katome_settings *s, const uint8_t *packed, uint64_t n_reads,
                        uint32_t read_len, const uint8_t *skip, katome_graph **out);

void katome_graph_free(katome_graph *g);

/* Stats<CollectionStats>::stats for the built graph (stats/collections.rs:137-168); host side */
int katome_graph_stats(const katome_graph *g, katome_stats *out);

/* message of the last failing call on this thread's process (what the reference panics with) */
const char *katome_last_error(void);

uint32_t katome_abi_version(void);

/* ---- host ingest alone (the step in front of the path; builder.rs:57-77,118-165) -------- */
typedef struct {
    uint64_t n_records;       /* records seen                                              */
    uint64_t n_reads;         /* accepted (all-ACGT) reads                                 */
    uint64_t read_bytes;      /* sum of accepted read lengths                              */
    uint64_t packed_bytes;
    uint64_t total_windows;   /* sum over reads of (len - k + 1)                           */
    uint32_t fixed_len;       /* != 0 iff every accepted read has this length              */
    uint32_t _pad;
    const uint8_t  *packed;   /* reads packed back to back, each starting on a byte        */
    const uint64_t *byte_off; /* [n_reads+1] start byte of each read in `packed`           */
    const uint32_t *len;      /* [n_reads] bases                                           */
} katome_reads;

int  katome_ingest_files(const katome_settings *s, const char *const *paths, size_t n_paths,
                         katome_reads **out);
void katome_reads_free(katome_reads *r);

/* ---- device-resident API -----------------------------------------------------------------
 * Used by bench.py (inputs already in HBM when the timed region starts) and by the multi-GPU
 * driver (one process per GPU; the k-mer exchange between extraction and insertion is an RCCL
 * all-to-all done by the caller).  All `d_` pointers are device pointers on settings.device;
 * `stream` is a hipStream_t (NULL = the default stream).  Calls are asynchronous on `stream`
 * unless they return a count.                                                              */
typedef struct katome_builder katome_builder;

int  katome_builder_create(const katome_settings *s, katome_builder **out);
void katome_builder_destroy(katome_builder *b);

/* optional per-phase timing with HIP events recorded on the caller's stream (bench.py's roofline
 * figures).  total_ms / launches have katome_phase_count() entries, named by katome_phase_name():
 * extract, region_order, insert, emit_edges, sort_edges, node_set, rank, labels, insert_tiles,
 * expand_tiles, expand_mid_tiles, first_seen_order.  Reading
 * synchronises the device and clears the record.                                            */
int  katome_builder_profile(katome_builder *b, int enable);
int  katome_builder_profile_read(katome_builder *b, double *total_ms, uint64_t *launches);
/* same, plus work[i] = elements processed by the launches of entry i (the per-kernel entries "k:...": keys of a sort pass,
 * slots of a table scan ...), so that launches of different sizes add up to one rate                              */
int  katome_builder_profile_read_work(katome_builder *b, double *total_ms, uint64_t *launches, uint64_t *work);
uint32_t katome_phase_count(void);
const char *katome_phase_name(uint32_t phase);

/* sizes seen by the last katome_dev_edges/finalize: out8 = {distinct tiles, tile-table slots, distinct stored
 * k-mers (one per strand pair when reverse_complement), k-mer-table slots, distinct mid tiles, mid-tile-table
 * slots, tile span, mid-tile span (0 = tiles expand straight into k-mers)}                          */
int  katome_builder_counts(katome_builder *b, uint64_t *out8);

/* u64 words per k-mer record for this k (1 or 2) */
uint32_t katome_record_words(uint32_t k);

/* k-mer extraction (compress_kmer / compress_kmer_with_rev_compl, compress.rs:18-48, over
 * read.windows(K), pt_graph.rs:294,310): one record per forward window, record =
 * packed k-mer (reverse_complement=0) or min(k-mer, reverse complement) (=1);
 * windows of skipped reads hold the invalid marker (all ones).
 * d_records: [n_reads*(read_len-k+1)][record_words] u64.                                   */
int katome_dev_extract_fixed(katome_builder *b, const uint8_t *d_packed, uint64_t n_reads,
                             uint32_t read_len, const uint8_t *d_skip, uint64_t *d_records,
                             void *stream);
/* variable-length reads: d_byte_off[n_reads+1], d_len[n_reads], d_win_prefix[n_reads+1]
 * (exclusive prefix sum of len-k+1); d_records: [total_windows][record_words]              */
int katome_dev_extract_var(katome_builder *b, const uint8_t *d_packed, uint64_t packed_bytes,
                           const uint64_t *d_byte_off, const uint32_t *d_len,
                           const uint64_t *d_win_prefix, uint64_t n_reads, uint64_t total_windows,
                           uint64_t *d_records, void *stream);

/* variable-length reads counted as tiles: with one span for the whole batch, read r yields (len_r-k+1)/span whole tiles
 * from its front (extract_var_tiles: d_tile_prefix[n_reads+1] = tiles before each read; records of
 * katome_tile_words(k, span) words, to katome_dev_insert_tiles) and its (len_r-k+1) mod span trailing windows as plain
 * k-mer records (extract_var_remainder: d_rest_prefix likewise; to katome_dev_insert, which also closes the batch --
 * call it even when total_rest is 0 on a FIRST_SEEN_ORDER builder).  d_win_prefix / total_windows as above.          */
int katome_dev_extract_var_tiles(katome_builder *b, const uint8_t *d_packed, uint64_t packed_bytes,
                                 const uint64_t *d_byte_off, const uint32_t *d_len, const uint64_t *d_tile_prefix,
                                 const uint64_t *d_win_prefix, uint64_t n_reads, uint64_t total_tiles,
                                 uint64_t total_windows, uint32_t span, uint64_t *d_records, void *stream);
int katome_dev_extract_var_remainder(katome_builder *b, const uint8_t *d_packed, uint64_t packed_bytes,
                                     const uint64_t *d_byte_off, const uint32_t *d_len, const uint64_t *d_rest_prefix,
                                     const uint64_t *d_win_prefix, uint64_t n_reads, uint64_t total_rest,
                                     uint64_t total_windows, uint32_t span, uint64_t *d_records, void *stream);

/* group records of `key_words` (1..3) u64 words by owner rank = mulhi(mix(key), n_parts) (stable; invalid records are
 * dropped); optional u32 values travel with their records (both d_values and d_values_out, or neither).
 * d_out: same size as d_records; h_counts[n_parts] receives the records per part (synchronises)       */
int katome_dev_partition(int device, const uint64_t *d_records, const uint32_t *d_values, uint64_t n_records,
                         uint32_t key_words, uint32_t n_parts, uint64_t *d_out, uint32_t *d_values_out,
                         uint64_t *h_counts, void *stream);
/* same, with the owner taken from the key's CORE instead of the whole key: the `core_bases` bases that end
 * `core_shift` bits above the key's low end, canonically (smaller of the core and its reverse complement):
 * owner = mulhi(mix(canonical core), n_parts).  The multi-GPU build routes k-mers with (core_shift 2, core k-2) --
 * the middle (k-2)-mer, which a k-mer shares with its reverse complement and with its source node's tail -- and
 * (k-1)-mer nodes with (0, k-2), so that every out-edge of a node, in either orientation's bookkeeping, lives on
 * the rank that owns the node (the HmGIR shape, hm_gir.rs:91-153: a node and its <= 4 out-edges in one place). */
int katome_dev_partition_core(int device, const uint64_t *d_records, const uint32_t *d_values, uint64_t n_records,
                              uint32_t key_words, uint32_t core_shift, uint32_t core_bases, uint32_t n_parts,
                              uint64_t *d_out, uint32_t *d_values_out, uint64_t *h_counts, void *stream);
/* host helper: the owner the two calls above compute for one key (core_bases 0 = whole key) */
uint32_t katome_key_owner(const uint64_t *key, uint32_t key_words, uint32_t core_shift, uint32_t core_bases,
                          uint32_t n_parts);

/* add_single_edge_fastaq (pt_graph.rs:172-198) for a batch: find-or-insert each record's
 * k-mer in the open-address table and add 1 to its weight (u32, wrapping).  Grows the table
 * when needed (synchronises).                                                              */
int katome_dev_insert(katome_builder *b, const uint64_t *d_records, uint64_t n_records, void *stream);
/* same with an explicit weight per record (BFCounter input, pt_graph.rs:201-213; and merging
 * pre-aggregated partial tables)                                                           */
int katome_dev_insert_weighted(katome_builder *b, const uint64_t *d_records, const uint32_t *d_weights,
                               uint64_t n_records, void *stream);

/* Tiled counting.  The insert kernel is bound by the rate of device-scope atomics, one per window.  For
 * fixed-length reads the windows can instead be counted in TILES -- the (k+span-1)-mers that cover `span`
 * consecutive windows, (read_len-k+1)/span per read -- and every distinct tile then adds its count to its
 * span k-mers at once (same sums as `weight += 1` per window, pt_graph.rs:186-191; ~span x fewer atomics).
 * katome_tile_span: the span the library would use for these reads (1 = plain counting); spans above 16 are
 * broken into mid tiles first (two levels of the same expansion).
 * Records of katome_dev_extract_tiles: [n_reads*((read_len-k+1)/span)][katome_tile_words(k, span)] u64 (whole tiles only).
 * Tiles are expanded into the k-mer table by katome_dev_edges / katome_dev_finalize; the multi-GPU
 * driver takes them out with katome_dev_expand_tiles as (k-mer, weight) records (library-owned, valid until
 * the next insert) and routes those to the k-mers' owners (katome_dev_insert_weighted).            */
uint32_t katome_tile_span(uint32_t k, uint32_t read_len);
uint32_t katome_tile_words(uint32_t k, uint32_t span);
/* Reads whose window count is not a multiple of a useful span (101 bp at k = 31: 71 windows) are counted as
 * *tiles from the front plus the windows that are left over*: katome_tile_plan picks the span with the fewest table
 * insertions per read (71 windows: 5 tiles of 14 + 1 single window = 6 instead of 71) and returns 1 if tiling
 * pays; katome_dev_extract_tiles then yields `tiles` records per read and katome_dev_extract_remainder the
 * `remainder` trailing windows of every read as plain k-mer records ([n_reads*remainder][katome_record_words(k)]),
 * to be passed to katome_dev_insert right after the tiles of the same batch.                          */
uint32_t katome_tile_plan(uint32_t k, uint32_t read_len, uint32_t *span, uint32_t *tiles, uint32_t *remainder);
/* same with tiles of at most max_tile_words u64 words (1: 31 bases, 2: 63, 3: 95 -- three-word tiles are what lets
 * k = 63 be tiled at all; katome_tile_plan = limit 3; the multi-GPU route, whose partition passes take one- and
 * two-word records, asks for 2)                                                                          */
uint32_t katome_tile_plan_limited(uint32_t k, uint32_t read_len, uint32_t max_tile_words, uint32_t *span,
                                  uint32_t *tiles, uint32_t *remainder);
int katome_dev_extract_remainder(katome_builder *b, const uint8_t *d_packed, uint64_t n_reads, uint32_t read_len,
                                 uint32_t span, const uint8_t *d_skip, uint64_t *d_records, void *stream);
int katome_dev_extract_tiles(katome_builder *b, const uint8_t *d_packed, uint64_t n_reads, uint32_t read_len,
                             uint32_t span, const uint8_t *d_skip, uint64_t *d_records, void *stream);
int katome_dev_insert_tiles(katome_builder *b, const uint64_t *d_records, uint64_t n_records, uint32_t span,
                            void *stream);
/* katome_dev_extract_tiles + katome_dev_insert_tiles in one call, without a record buffer of the caller's: one batch of reads
 * of one length counted as tiles (add_read_fastaq's windows, pt_graph.rs:277-315, `span` at a time).  When the builder keeps tile
 * records aside to count them by sorting (by packed key, tiles of two words) and no read is skipped (d_skip == NULL), the records
 * are written where they are kept.  The windows behind a read's last whole tile still go through
 * katome_dev_extract_remainder + katome_dev_insert.                                                        */
int katome_dev_count_tiles(katome_builder *b, const uint8_t *d_packed, uint64_t n_reads, uint32_t read_len,
                           uint32_t span, const uint8_t *d_skip, void *stream);
int katome_dev_expand_tiles(katome_builder *b, uint64_t **d_keys, uint32_t **d_weights, uint64_t *n_records,
                            void *stream);

/* Clean::remove_weak_edges(threshold) for PtGraph (pruner.rs:84-93): keep the edges with weight >= threshold,
 * then drop the vertices left without neighbours.  Call before katome_dev_edges / katome_dev_finalize: the
 * filter is applied when the edges are read out of the table, so the sort and the node numbering only see
 * the surviving edges (and the nodes are exactly their endpoints).
 * FIRST_SEEN_ORDER builders instead number the whole graph first and then remove as petgraph's retain_edges /
 * retain_nodes do (indices visited in descending order, rejected ones swap_removed), so the result keeps the
 * reference's numbering; on such a builder the call is also accepted AFTER katome_dev_finalize (e.g. after
 * katome_dev_remove_dead_paths, the order of asm/basic_assembler.rs:58-66) and then acts at once.       */
int katome_dev_remove_weak_edges(katome_builder *b, uint32_t threshold, void *stream);

/* number of distinct keys in the table so far (synchronises) */
int katome_dev_table_count(katome_builder *b, uint64_t *out, void *stream);

/* device-resident graph; arrays owned by the builder until destroy / next finalize */
typedef struct {
    uint64_t n_nodes, n_edges;
    uint32_t key_words, label_stride;
    uint64_t *d_edge_key;     /* [n_edges][key_words], ascending */
    uint32_t *d_edge_weight;
    uint64_t *d_edge_src, *d_edge_dst;
    uint8_t  *d_edge_label;
    uint64_t *d_node_key;     /* [n_nodes][key_words]: sources ascending, then out-edge-less nodes ascending */
    uint32_t *d_edge_age;     /* see katome_graph.edge_age; NULL while age == index */
} katome_dev_graph;

/* Table -> distinct oriented edges, sorted by packed k-mer (both strands when
 * reverse_complement, pt_graph.rs:282-308); then node numbering, endpoints and labels
 * (the PtGraph::create post-pass, pt_graph.rs:339-343 -> kmer_to_edge compress.rs:231-233). */
int katome_dev_finalize(katome_builder *b, katome_dev_graph *out, void *stream);

/* Prunable::remove_dead_paths for PtGraph (pruner.rs:36-82; Externals 165-195, remove_paths 199-217,
 * check_dead_path 229-257), in place on the finalized graph of a FIRST_SEEN_ORDER builder: call after
 * katome_dev_finalize; `graph` is updated (counts shrink, labels are rewritten).  Everything runs on the device:
 * the walks, the degree bookkeeping, the two swap_remove replays that fix petgraph's re-numbering (edges: a scan
 * + pointer jumping; nodes: the vacated tail positions settled in rounds) and the array moves.  Only a pass whose
 * node moves chain further than the device form follows is replayed sequentially on the host (see prune.hip).  */
typedef struct {
    uint64_t passes;                 /* iterations of the reference's outer loop, the last (empty) one included */
    uint64_t walks, dead_walks;      /* walks started from vertices without incoming edges / walks found dead   */
    uint64_t marked;                 /* edge indices collected, duplicates counted                              */
    uint64_t removed_edges, removed_by_duplicates, removed_nodes;
    double   host_ms;                /* time in sequential host replays (0 unless a pass fell back to them)     */
    double   total_ms;
} katome_prune_stats;
int katome_dev_remove_dead_paths(katome_builder *b, katome_dev_graph *graph, katome_prune_stats *stats, void *stream);

/* the finalized graph as it stands now (after katome_dev_remove_dead_paths / katome_dev_remove_weak_edges) */
int katome_dev_current_graph(katome_builder *b, katome_dev_graph *out);

/* Standardizable for PtGraph (standardizer.rs:41-128), the two weight passes assemble_with_graph runs between its
 * prunings (asm/basic_assembler.rs:63-70), in place on the finalized graph as it stands:
 *  - standardize_contigs (72-122): every contig -- an out-edge of an ambiguous vertex (pt_graph.rs:54-62) followed while
 *    the vertex reached has one out-edge and is not ambiguous -- gets the rounded mean of its weights; no re-numbering;
 *  - standardize_edges (42-70): weights scaled by (original_genome_length - k) / (sum of weights - sum of the weights
 *    under `threshold`), rounded, then remove_weak_edges(1) (same numbering rules as katome_dev_remove_weak_edges).
 * With katome_dev_remove_dead_paths / katome_dev_remove_weak_edges this covers every stage of the reference up to
 * `collapse`, index for index on a FIRST_SEEN_ORDER builder (fetch the arrays with katome_dev_current_graph).      */
int katome_dev_standardize_contigs(katome_builder *b, void *stream);
int katome_dev_standardize_edges(katome_builder *b, uint64_t original_genome_length, uint32_t threshold, void *stream);

/* host result of katome_shrink_files / katome_shrink_packed: the build (with the pruning the flags ask for) followed by
 * Shrinkable::shrink, see katome_dev_shrink below for what the arrays mean.  Owned by the library until katome_contigs_free */
typedef struct {
    uint64_t n_nodes, n_edges, label_bytes, read_bytes;
    uint32_t k, key_words;
    const uint64_t *edge_src, *edge_dst;       /* [n_edges] ids into node_key                                  */
    const uint32_t *edge_weight, *edge_kmers;  /* weight of the path's first k-mer; k-mers merged into the edge  */
    const uint64_t *edge_label_off;            /* [n_edges + 1] byte offsets into edge_label                    */
    const uint8_t  *edge_label;                /* compress_edge format, edge i = bytes [off[i], off[i+1])        */
    const uint64_t *node_key;                  /* [n_nodes][key_words] packed (k-1)-mers                         */
} katome_contigs;
int  katome_shrink_files(const katome_settings *s, const char *const *paths, size_t n_paths, katome_contigs **out);
int  katome_shrink_packed(const katome_settings *s, const uint8_t *packed, uint64_t n_reads, uint32_t read_len,
                          const uint8_t *skip, katome_contigs **out);
void katome_contigs_free(katome_contigs *c);

/* Shrinkable::shrink for PtGraph (shrinker.rs:165-209; labels merged as EdgeSlice::merge does, slices.rs:23-34) on
 * the finalized graph as it stands: every maximal straight path (inner vertices with exactly one edge in and one out)
 * becomes ONE edge spelling the whole path, with the weight of the path's first edge (shrinker.rs:181,200); the inner
 * vertices disappear (remove_single_vertices, shrinker.rs:172).  The graph of the builder is left untouched; the
 * result is a separate set of device arrays owned by the builder (valid until the next call or destroy):
 *   d_edge_label: the paths in compress_edge format (compress.rs:250-271: [pad][packed bases, left-aligned]), edge i
 *   at bytes [d_edge_label_off[i], d_edge_label_off[i+1]); d_edge_kmers[i] = k-mers merged into edge i.
 * Two forms (katome_dev_shrink_mode):
 *  - KATOME_SHRINK_EXACT: the reference's own result -- where ShrinkTraverse (shrinker.rs:62-135) cuts paths on tangled graphs
 *    and cycles, and the edge / node indices petgraph's swap_removes and add_edges leave (shrink_single_path 178-209,
 *    remove_single_vertices) -- index for index.  That order is a sequential depth-first traversal interleaved with the
 *    mutation, so it is computed on one host core over petgraph's own layout (csrc/shrink_exact.h; *host_ms reports it);
 *    adjacency order (edge ages) before and weights, k-mer counts and labels after are the device's.
 *  - KATOME_SHRINK_FAST: everything on the device, traversal-free: the same SET of merged edges wherever the reference's
 *    traversal starts from a vertex without incoming edges, cycles of inner vertices cut at their smallest vertex; numbering
 *    = edges in the order of their first k-mer in the input graph, vertices in their input order.
 *  - KATOME_SHRINK_AUTO (katome_dev_shrink): exact on a FIRST_SEEN_ORDER builder, fast otherwise. */
#define KATOME_SHRINK_AUTO  0u
#define KATOME_SHRINK_FAST  1u
#define KATOME_SHRINK_EXACT 2u
typedef struct {
    uint64_t  n_nodes, n_edges, label_bytes;
    uint32_t  key_words, _pad;
    uint64_t *d_edge_src, *d_edge_dst;
    uint32_t *d_edge_weight, *d_edge_kmers;
    uint64_t *d_edge_label_off;
    uint8_t  *d_edge_label;
    uint64_t *d_node_key;
} katome_dev_contigs;
int katome_dev_shrink(katome_builder *b, katome_dev_contigs *out, void *stream);
int katome_dev_shrink_mode(katome_builder *b, uint32_t mode, katome_dev_contigs *out, double *host_ms, void *stream);

/* first half of finalize only: sorted distinct edges (key, weight); used by the multi-GPU
 * driver, which resolves node ids across ranks itself                                      */
int katome_dev_edges(katome_builder *b, uint64_t **d_edge_key, uint32_t **d_edge_weight,
                     uint64_t *n_edges, void *stream);

/* The library keeps freed device blocks for reuse (hipMalloc/hipFree of multi-GiB buffers are slow);
 * this hands them back to the driver.                                                         */
int katome_dev_release_cache(int device);
/* out[0] = bytes the library holds from the driver on `device` (all segments), out[1] = how many of them are free
 * (cached), out[2] = blocks handed out and not yet returned.  After katome_dev_release_cache() out[0] is what live
 * builders / results still pin -- 0 when everything was closed.                                                   */
int katome_dev_cache_stats(int device, uint64_t out[3]);

/* ---- multi-GPU: the sharded build, one rank per GPU -------------------------------------------------------
 * The reference is one sequential loop (builder.rs:152-160); what shards is the read set.  Reads are split contiguously by
 * index over the ranks; every rank extracts its own records and routes each to its owner rank; owners count; the graph
 * is numbered across ranks.  Inside katome_build_* (settings.n_devices) the ranks are host threads of the calling process;
 * a job that runs one PROCESS per GPU (bench.py under a launcher) creates a communicator per process and drives
 * katome_dist_* itself.                                                                                       */
typedef struct katome_comm katome_comm;
#define KATOME_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId), every rank gets it out of band and joins (ncclCommInitRank) */
int katome_comm_unique_id(uint8_t *id /* [KATOME_COMM_ID_BYTES] */);
int katome_comm_create_rccl(const uint8_t *id, int rank, int world, int device, katome_comm **out);
/* the caller moves the bytes (tests: torch.distributed/gloo): element counts and offsets per peer, host or device
 * buffers (on_device); op: 0 sum, 1 max, 2 min.  Non-zero return = failure.                                 */
typedef struct {
    void *user;
    int (*alltoallv)(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_cnt, void *recv,
                     const uint64_t *recv_off, const uint64_t *recv_cnt, uint64_t elem_bytes, int on_device);
    int (*allreduce_u64)(void *user, uint64_t *vals, uint64_t n, int op);
} katome_comm_callbacks;
int katome_comm_create_callbacks(const katome_comm_callbacks *cb, int rank, int world, int device, katome_comm **out);
void katome_comm_destroy(katome_comm *c);
int katome_comm_rank(const katome_comm *c);
int katome_comm_world(const katome_comm *c);
const char *katome_comm_kind(const katome_comm *c);               /* "rccl", "local", "callbacks" */
/* a single message above this many bytes travels in rounds (default 1 GiB) */
int katome_comm_set_max_message_bytes(katome_comm *c, uint64_t bytes);
int katome_comm_allreduce_u64(katome_comm *c, uint64_t *vals, uint64_t n, int op);
/* variable all-to-all: `send` holds send_cnt[p] elements for each peer p, in peer order; recv_cnt[p] is filled in and
 * `recv` (room for recv_capacity elements) receives them grouped by source.  Collective.                   */
int katome_comm_exchange(katome_comm *c, const void *send, const uint64_t *send_cnt, void *recv, uint64_t recv_capacity,
                         uint64_t *recv_cnt, uint64_t elem_bytes, int on_device, void *stream);

typedef struct katome_dist_builder katome_dist_builder;
/* one rank's share of the graph, device arrays owned by the builder */
typedef struct {
    uint64_t n_edges, n_nodes;            /* on this rank: its edges are the out-edges of the nodes it owns ...      */
    uint64_t total_edges, total_nodes;    /* ... of the whole graph                                                  */
    uint64_t node_base;                   /* by packed key: global id of this rank's node i = node_base + i           */
    uint32_t key_words, label_stride;
    uint64_t *d_edge_key;                 /* [n_edges][key_words] ascending                                          */
    uint32_t *d_edge_weight;
    uint64_t *d_edge_src, *d_edge_dst;    /* GLOBAL node ids                                                         */
    uint8_t  *d_edge_label;
    uint64_t *d_node_key;                 /* [n_nodes][key_words]                                                    */
    uint64_t *d_edge_id, *d_node_id;      /* FIRST_SEEN_ORDER: the reference's (petgraph) index of every edge / node of
                                             this rank; NULL by packed key                                           */
    uint64_t *d_edge_age;                 /* after katome_dist_remove_dead_paths: the index each edge had when it was built
                                             (= its place in petgraph's adjacency lists, see katome_graph.edge_age); else NULL */
} katome_dist_graph;
/* settings as for katome_builder_create (settings.device = this rank's GPU; table_slots_hint for the WHOLE build);
 * the communicator stays the caller's.  Every call below is collective: all ranks make it, in the same order.        */
int  katome_dist_create(const katome_settings *s, katome_comm *comm, katome_dist_builder **out);
void katome_dist_destroy(katome_dist_builder *d);
/* this rank's reads, device pointers: reads [first_read, first_read + n_reads) of the whole input in input order (the
 * reference's numbering needs the global index), fixed length, in batches of batch_reads (0 = default).  May be called
 * again with the reads that follow.                                                                           */
int  katome_dist_add_reads(katome_dist_builder *d, const uint8_t *d_packed, uint64_t first_read, uint64_t n_reads,
                           uint32_t read_len, const uint8_t *d_skip, uint64_t batch_reads, void *stream);
/* Clean::remove_weak_edges(threshold) when the edges are read out (by packed key only; see katome_dev_remove_weak_edges) */
int  katome_dist_remove_weak_edges(katome_dist_builder *d, uint32_t threshold);
int  katome_dist_finalize(katome_dist_builder *d, katome_dist_graph *out, void *stream);
/* FIRST_SEEN_ORDER: bring the whole graph to rank `root` in the reference's index order and hand it to a single-GPU
 * builder there (*root_builder; NULL on the other ranks; owned by `d`), on which katome_dev_remove_dead_paths,
 * katome_dev_remove_weak_edges, katome_dev_standardize_*, katome_dev_shrink and katome_dev_current_graph work as after
 * katome_dev_finalize.  The pruning of BASELINE config 5 runs this way: its walks and swap-removes depend on the global
 * numbering; the whole graph has to fit the root's HBM (< 2^32 edges).                           */
int  katome_dist_gather(katome_dist_builder *d, int root, katome_builder **root_builder, void *stream);
/* Prunable::remove_dead_paths (pruner.rs:36-82) on the SHARDED graph of a finalized FIRST_SEEN_ORDER build, no gather: walkers
 * hop from owner to owner along first out-edges (one all-to-all per step, < 2k steps), marked (position, count) pairs go to
 * rank 0, which replays petgraph's swap_removes on 64-bit positions and answers where every candidate edge / node ends up
 * (katome_amd/csrc/dist_prune.hip).  The graph may hold more than 2^32 edges (a rank's share may not).  Afterwards `out`
 * describes this rank's share of the pruned graph: d_edge_id / d_node_id are the indices the reference's PtGraph would hold,
 * d_edge_age the edges' ages.  Collective.                                                                              */
int  katome_dist_remove_dead_paths(katome_dist_builder *d, katome_dist_graph *out, katome_prune_stats *stats, void *stream);
/* this rank's share of the graph as it stands (after katome_dist_finalize / katome_dist_remove_dead_paths) */
int  katome_dist_current_graph(katome_dist_builder *d, katome_dist_graph *out);
/* the rank's single-GPU builder underneath (per-phase kernel timing: katome_builder_profile*) */
katome_builder *katome_dist_inner(katome_dist_builder *d);
/* which records travel in this build: "local" (every rank counts its own reads, distinct k-mers routed: up to two ranks), "tiles"
 * (tiles, mid tiles and k-mer records routed level by level: three ranks and more), "supermers" (one exchange of 16-byte supermer
 * records before any counting: KATOME_DIST_ROUTE=supermers, by packed key, k <= 31); KATOME_DIST_ROUTE overrides the choice */
const char *katome_dist_route(const katome_dist_builder *d);
/* exchange accounting since the last read: katome_dist_exchange_count() phases named by katome_dist_exchange_name(),
 * out[4*i..] = {calls, bytes that left this rank, largest single (rank -> peer) message, microseconds inside the exchange} */
uint32_t katome_dist_exchange_count(void);
const char *katome_dist_exchange_name(uint32_t phase);
int  katome_dist_exchange_read(katome_dist_builder *d, uint64_t *out);
/* contiguous shard of `total_reads` for `rank` (starts are multiples of 64 reads: 16-byte aligned packed rows) */
void katome_shard_range(uint64_t total_reads, uint32_t world, uint32_t rank, uint64_t *first, uint64_t *count);

/* ---- device primitives the finalize is built from (exported for the multi-GPU driver and
 * for unit tests; each is a hand-written HIP kernel set) -------------------------------- */
/* LSD radix sort of n keys of `key_words` u64 words on bits [0, key_bits); optional u32
 * values.  d_keys/d_vals are sorted in place (a temporary of equal size is allocated).     */
int katome_dev_sort(int device, uint64_t *d_keys, uint32_t *d_vals, uint64_t n, uint32_t key_words,
                    uint32_t key_bits, void *stream);
/* remove_paths' edge removals (pruner.rs:199-217: the collected indices sorted descending, Graph::remove_edge =
 * swap_remove, an index listed twice removes what was moved in) as katome_dev_remove_dead_paths runs them: d_pos[u]
 * ascending marked positions, d_mult[u] how often each is listed, of n_edges edges.  -> d_victims (identity of every
 * removed edge, in removal order; room for sum(d_mult)), the moves d_move_to[i] <- d_move_from[i] that fill the marked
 * positions below the new count (room for u), counts = {removed, moves, edges left, removals owed to repeats}      */
int katome_dev_replay_edge_removals(int device, const uint32_t *d_pos, const uint32_t *d_mult, uint64_t u, uint64_t n_edges,
                                    uint32_t *d_victims, uint32_t *d_move_to, uint32_t *d_move_from, uint64_t *counts,
                                    void *stream);
/* remove_single_node after every removed edge (pruner.rs:206-225; Graph::remove_node = swap_remove): d_die[2t],
 * d_die[2t+1] = the source / target that edge removal t leaves without edges (0xFFFFFFFF: stays); when both go, the
 * one at the larger current index goes first.  -> the moves d_move_to[i] <- d_move_from[i] of the nodes that end up
 * re-numbered (room for as many as die), counts = {moves, nodes left, 1 if the device form gave up (chains of moves
 * longer than it follows: katome_dev_remove_dead_paths then runs the sequential replay on the host; nothing is written)} */
int katome_dev_replay_node_removals(int device, const uint32_t *d_die, uint64_t m, uint64_t n_nodes, uint32_t *d_move_to,
                                    uint32_t *d_move_from, uint64_t *counts, void *stream);
/* the same two replays on 64-bit positions (0xFFFFFFFFFFFFFFFF: stays): what katome_dist_remove_dead_paths runs on rank 0 for
 * a graph sharded over several GPUs, which may hold more than 2^32 edges (BASELINE config 5: 1.1e10).  Only the marked
 * entries and the tail positions that disappear are touched, so n_edges / n_nodes are not bounded by any buffer.         */
int katome_dev_replay_edge_removals64(int device, const uint64_t *d_pos, const uint32_t *d_mult, uint64_t u, uint64_t n_edges,
                                      uint64_t *d_victims, uint64_t *d_move_to, uint64_t *d_move_from, uint64_t *counts,
                                      void *stream);
int katome_dev_replay_node_removals64(int device, const uint64_t *d_die, uint64_t m, uint64_t n_nodes, uint64_t *d_move_to,
                                      uint64_t *d_move_from, uint64_t *counts, void *stream);
/* exclusive prefix sums of m u32 counts as u64: d_offs[i] = counts[0] + .. + counts[i-1], d_offs[m] = the total   */
int katome_dev_scan_counts(int device, const uint32_t *d_counts, uint64_t m, uint64_t *d_offs, void *stream);
/* in-place unique of sorted keys; returns the new count (synchronises)                     */
int katome_dev_unique(int device, uint64_t *d_keys, uint64_t n, uint32_t key_words, uint64_t *n_out,
                      void *stream);
/* rank of each query key in a sorted unique key array (every query must be present; absent
 * keys yield UINT64_MAX)                                                                   */
int katome_dev_rank(int device, const uint64_t *d_sorted, uint64_t n_sorted, uint32_t key_words,
                    uint32_t key_bits, const uint64_t *d_queries, uint64_t n_queries,
                    uint64_t *d_rank_out, void *stream);
/* node numbering of a sorted distinct edge list (what katome_dev_finalize does): d_node_key receives the
 * node keys (capacity [2*n_edges][key_words]: the sources in ascending order, then the out-edge-less
 * targets in ascending order), d_edge_src/d_edge_dst [n_edges] the positions of each edge's endpoints in
 * it; *n_nodes the node count (synchronises)                                                    */
int katome_dev_node_ids(int device, const uint64_t *d_edge_key, uint64_t n_edges, uint32_t k, uint64_t *d_node_key,
                        uint64_t *d_edge_src, uint64_t *d_edge_dst, uint64_t *n_nodes, void *stream);
/* derive (k-1)-mer endpoint keys of each edge: d_src/d_dst [n][key_words]                  */
/* first half of katome_dev_node_ids: the distinct source (k-1)-mers of sorted edges, ascending (d_node_key:
 * room for n_edges keys), and each edge's position among them                                          */
int katome_dev_source_ids(int device, const uint64_t *d_edge_key, uint64_t n_edges, uint32_t k,
                          uint64_t *d_node_key, uint64_t *d_edge_src, uint64_t *n_sources, void *stream);
/* source / target (k-1)-mer of every edge (either output may be NULL) */
int katome_dev_endpoints(int device, const uint64_t *d_edge_key, uint64_t n, uint32_t k,
                         uint64_t *d_src_key, uint64_t *d_dst_key, void *stream);
/* compress_edge-format labels (compress.rs:250-271) of packed k-mers                       */
int katome_dev_labels(int device, const uint64_t *d_edge_key, uint64_t n, uint32_t k,
                      uint8_t *d_label, void *stream);

/* ---- synthetic workload generator (bench/tests; DESIGN.md "Synthetic workload") ---------- */
int katome_dev_synth_reads(int device, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                           uint64_t genome_len, double err_rate, uint32_t n_inject_percent,
                           uint8_t *d_packed, uint8_t *d_skip, void *stream);

#ifdef __cplusplus
}
#endif
#endif
