/*
 * katome_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of fuine/katome's `build` stage (k-mer extraction + de Bruijn
 * graph construction).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; nothing under katome_amd/ links, imports
 * or calls it.  See oracle/README.md for the pin status.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout, src/katome/...).
 */
#ifndef KATOME_ORACLE_H
#define KATOME_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- prelude.rs:21-43 ------------------------------------------------------------ */
void   ko_set_global_k_sizes(size_t k);       /* prelude.rs:34-43 (asserts k > 1) */
size_t ko_k_size(void);
size_t ko_k1_size(void);
size_t ko_compressed_k1_size(void);

/* ---- compress.rs ----------------------------------------------------------------- */
uint8_t ko_encode_fasta_symbol(uint8_t symbol, uint8_t carrier);              /* 347-378 */
size_t  ko_compress_node(const uint8_t *slice, size_t len, uint8_t *out);     /* 55-73   */
size_t  ko_compress_kmer(const uint8_t *kmer, size_t len, uint8_t *out);      /* 18-28   */
size_t  ko_compress_kmer_with_rev_compl(const uint8_t *kmer, size_t len,
                                        uint8_t *out, uint8_t *rev);          /* 34-48   */
void    ko_reverse_compressed_node(const uint8_t *compr, size_t n,
                                   size_t remainder_size, uint8_t *out);      /* 153-169 */
void    ko_shift_left_bit_array(uint8_t *buf, size_t n, size_t shift);        /* 403-419 */
void    ko_shift_right_bit_array(uint8_t *buf, size_t n, size_t shift);       /* 426-442 */
size_t  ko_compress_edge(const uint8_t *edge, size_t len, uint8_t *out);      /* 250-271 */
size_t  ko_decompress_edge(const uint8_t *edge, size_t n, uint8_t *out);      /* 283-293 */
size_t  ko_decompress_node(const uint8_t *node, size_t n, uint8_t *out);      /* 78-85   */
size_t  ko_decompress_kmer(const uint8_t *kmer, size_t n, uint8_t *out);      /* 90-97   */
size_t  ko_kmer_to_edge(const uint8_t *kmer, size_t n, uint8_t *out);         /* 231-233 */
size_t  ko_add_char_to_edge(const uint8_t *edge, size_t n, uint8_t chr,
                            uint8_t *out);                                    /* 205-227 */
size_t  ko_change_last_char_in_edge(const uint8_t *edge, size_t n, uint8_t to,
                                    uint8_t *out);                            /* 193-201 */
size_t  ko_extend_edge(const uint8_t *edge, size_t n, const uint8_t *with,
                       size_t with_len, uint8_t *out);                        /* 174-189 */
uint8_t ko_change_char_in_chunk(uint8_t chunk, size_t offset, uint8_t to);    /* 109-115 */
char    ko_decompress_char(uint8_t chunk, size_t padding);                    /* 304-314 */
void    ko_decode_compressed_chunk(uint8_t chunk, uint8_t out[4]);            /* 381-397 */

/* ---- stats/collections.rs:38-57,137-168 ------------------------------------------ */
typedef struct {
    uint64_t node_count, edge_count;
    uint32_t max_edge_weight;
    double   avg_edge_weight;
    uint64_t max_in_degree, max_out_degree;
    double   avg_out_degree;
    uint64_t incoming_vert_count, outgoing_vert_count;
} ko_stats;

/* ---- result of PtGraph::create (pt_graph.rs:333-345) ------------------------------ */
typedef struct {
    uint64_t  n_nodes, n_edges, read_bytes;
    uint64_t *edge_src, *edge_dst;   /* petgraph first-seen node ids              */
    uint32_t *edge_weight;
    uint64_t *edge_slot;             /* EdgeSlice offset into SEQUENCES            */
    uint8_t  *edge_label;            /* [n_edges][label_stride] compress_edge fmt  */
    uint32_t  label_stride;          /* 1 + ceil(k/4)                              */
    uint64_t  n_sequences;           /* len(SEQUENCES) incl. scratch slot 0        */
    ko_stats  stats;
    /* HmGIR/HsGIR observable (stats/collections.rs:170-208): same sets, counted by the
     * GIR restatement (hm_gir.rs:91-153) run beside the graph build.                 */
    uint64_t  gir_node_count, gir_edge_count;
    /* after a 's' (shrink) stage: every edge's whole sequence as ASCII, edge e = edge_seq[edge_seq_off[e] ..
     * edge_seq_off[e+1]); edge_label then only holds meaningful bytes for edges that were never merged        */
    uint64_t *edge_seq_off;
    uint8_t  *edge_seq;
    /* after a 'C' (collapse) stage: the serialized contigs in the order collapse() returns them (collapser.rs:29-82);
     * the graph itself is then what collapse left of it (normally nothing) */
    uint64_t  n_contigs;
    uint64_t *contig_off;
    uint8_t  *contig_seq;
} ko_graph;

/* error codes mirror the reference's panics */
enum {
    KO_OK = 0,
    KO_E_PATH = -1,        /* builder.rs:62  "Coulndt resolve path"   */
    KO_E_IS_DIR = -2,      /* builder.rs:67  "is a directory"         */
    KO_E_NOT_EXIST = -3,   /* builder.rs:71  "does not exist"         */
    KO_E_OPEN = -4,        /* builder.rs:124,148 "Couldn't open all files" */
    KO_E_PARSE = -5,       /* builder.rs:128,153 record unwrap        */
    KO_E_SHORT_READ = -6,  /* pt_graph.rs:278 "Read is too short!"    */
    KO_E_ARG = -7
};

/* file_type: 0 Fasta, 1 Fastq (config.rs:5-14); BFCounter not restated (unpinned) */
int  ko_build_files(const char *const *paths, size_t n_paths, int file_type,
                    int reverse_complement, size_t k, int with_gir, ko_graph **out);
/* InputFileType::BFCounter (builder.rs:79-115; pt_graph.rs:201-213,317-330): `kmer\tweight` lines, lines below the
 * threshold dropped, every kept line one edge (two with reverse_complement), duplicates kept as parallel edges */
int  ko_build_bfc(const char *const *paths, size_t n_paths, int reverse_complement, uint32_t threshold, size_t k,
                  ko_graph **out);
/* reads given as fixed-length ASCII rows (the synthetic workloads); the ACGT filter
 * of builder.rs:155-157 is applied exactly as for file input                        */
int  ko_build_ascii(const uint8_t *reads, size_t n_reads, size_t read_len,
                    int reverse_complement, size_t k, int with_gir, ko_graph **out);
/* when on, every ko_build_* runs Prunable::remove_dead_paths (pruner.rs:36-82; restated with petgraph 0.4.13's
 * swap-remove index semantics) on the finished PtGraph before the result arrays are read out */
void ko_set_prune_dead_paths(int on);
uint64_t ko_last_prune_passes(void);
/* general form: `stages` is applied in order to the finished PtGraph of every ko_build_*: 'd' = remove_dead_paths,
 * 'C' = Collapsable::collapse (collapser.rs:29-273, over a restated petgraph tarjan_scc),
 * 'w' = Clean::remove_weak_edges(weak_threshold) (pruner.rs:84-93, over petgraph's retain_edges / retain_nodes:
 * indices visited in descending order, rejected ones swap_removed), 's' = Shrinkable::shrink (shrinker.rs:38-209,
 * with EdgeSlice::merge slices.rs:23-34); "" = none */
void ko_set_post_build(const char *stages, uint32_t weak_threshold);
/* further stages: 'c' = Standardizable::standardize_contigs (standardizer.rs:72-122), 'e' = standardize_edges
 * (standardizer.rs:42-70) with original_genome_length set here, k = the build's k and threshold = weak_threshold */
void ko_set_genome_length(uint64_t original_genome_length);
/* PtGraph::from_edges + shrink on a hand-made graph (the in-file tests of shrinker.rs:237-488): slot i of SEQUENCES =
 * compress_edge(slot_ascii[i]) (slot 0 unused); edge j = (src[j], dst[j], (EdgeSlice(slot[j]), w[j])).  Result: endpoints,
 * weights and edge_seq of the shrunk graph (no fixed-stride labels) */
int  ko_shrink_from_edges(const uint64_t *src, const uint64_t *dst, const uint64_t *slot, const uint32_t *w, size_t n_edges,
                          const char *const *slot_ascii, size_t n_slots, size_t k, ko_graph **out);
/* general form: n_nodes add_node calls first (pruner.rs:277-285 style), any stage string ('s','d','w','v' = remove_single_vertices) */
int  ko_run_from_edges(size_t n_nodes, const uint64_t *src, const uint64_t *dst, const uint64_t *slot, const uint32_t *w, size_t n_edges,
                       const char *const *slot_ascii, size_t n_slots, const char *stages, uint32_t threshold, size_t k, ko_graph **out);
void ko_graph_free(ko_graph *g);
const char *ko_last_error(void);

/* FASTQ/FASTA record scan alone (restated bio 0.10.0 reader, see .c): returns accepted
 * reads concatenated as ASCII with offsets; used to cross-check the product's ingest */
typedef struct {
    uint64_t  n_records, n_accepted, read_bytes;
    uint8_t  *seq;          /* accepted reads, concatenated */
    uint64_t *off;          /* [n_accepted+1] */
} ko_reads;
int  ko_scan_files(const char *const *paths, size_t n_paths, int file_type, ko_reads **out);
void ko_reads_free(ko_reads *r);

/* ---- synthetic workload generator (our definition, not the reference's) */
uint64_t ko_splitmix64(uint64_t x);
/* writes n_reads rows of read_len ASCII bases; reads chosen for N-injection get one 'N' */
void ko_synth_reads(uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                    uint64_t genome_len, double err_rate, uint32_t n_inject_percent,
                    uint8_t *out_ascii);

#ifdef __cplusplus
}
#endif
#endif
