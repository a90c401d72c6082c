/*
 * katome_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's `build` stage, written from a reading of the
 * Rust sources (no Rust toolchain exists in this environment, so the reference cannot be
 * compiled or run: the oracle is pinned instead by every constant and known-answer vector
 * the reference's own tests hold for this path -- see tests/test_oracle_*.py and
 * tests/golden/pinned.json).
 *
 * Third-party arithmetic that is NOT under the reference checkout and is restated from
 * its published behaviour (versions from Cargo.lock):
 *   - bio 0.10.0   io::fastq / io::fasta record readers (record framing, `seq()` =
 *                  right-trimmed line).  Pinned only through read_bytes / counts of the
 *                  three fixtures (tests/build.rs:27-28); CRLF / blank-line / multi-line
 *                  FASTQ behaviour is "parity unpinned".
 *   - petgraph 0.4.13  Graph::add_node/add_edge/find_edge adjacency-list semantics
 *                  (new edge becomes the head of both endpoint lists).
 *   - metrohash 0.2.0  only decides HashMap iteration order, which nothing on this path
 *                  observes; the oracle uses FNV-1a instead.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this file.
 */
#define _GNU_SOURCE
#include "katome_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#define CHARS_PER_CARRIER 4 /* compress.rs:11 */

static size_t K_SIZE = 40, K1_SIZE = 39, COMPRESSED_K1_SIZE = 10; /* prelude.rs:21-25 */
static char g_err[512];

const char *ko_last_error(void) { return g_err; }

static size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

/* prelude.rs:34-43 */
void ko_set_global_k_sizes(size_t k)
{
    if (k <= 1) { fprintf(stderr, "assertion failed: k_size > 1\n"); abort(); }
    K_SIZE = k;
    K1_SIZE = k - 1;
    COMPRESSED_K1_SIZE = ceil_div(K1_SIZE, CHARS_PER_CARRIER);
}
size_t ko_k_size(void) { return K_SIZE; }
size_t ko_k1_size(void) { return K1_SIZE; }
size_t ko_compressed_k1_size(void) { return COMPRESSED_K1_SIZE; }

/* ================================ compress.rs ===================================== */

/* compress.rs:347-378 -- the bit trick on the ASCII code, kept verbatim in meaning:
 * symbol -= 'A'; symbol >>= 1; first = (~C) & D ; second = C | A                     */
uint8_t ko_encode_fasta_symbol(uint8_t symbol, uint8_t carrier)
{
    carrier = (uint8_t)(carrier << 2);
    symbol = (uint8_t)(symbol - 'A');
    symbol >>= 1;
    uint8_t c_masked = (symbol & 2) >> 1;
    uint8_t a_masked = (symbol & 8) >> 3;
    uint8_t d_masked = symbol & 1;
    uint8_t first_bit = (c_masked ^ 1) & d_masked;
    uint8_t second_bit = c_masked | a_masked;
    return (uint8_t)(carrier | ((second_bit << 1) | first_bit));
}

/* compress.rs:55-73: 4 symbols per byte, MSB first; the last byte is left-aligned */
size_t ko_compress_node(const uint8_t *slice, size_t len, uint8_t *out)
{
    size_t n = 0;
    for (size_t i = 0; i < len; i += CHARS_PER_CARRIER) {
        uint8_t carrier = 0;
        size_t chunk = len - i < CHARS_PER_CARRIER ? len - i : CHARS_PER_CARRIER;
        for (size_t j = 0; j < chunk; ++j) carrier = ko_encode_fasta_symbol(slice[i + j], carrier);
        out[n++] = carrier;
    }
    if (len) {
        size_t last = len % CHARS_PER_CARRIER ? len % CHARS_PER_CARRIER : CHARS_PER_CARRIER;
        size_t l = CHARS_PER_CARRIER - last;
        if (l != 0) out[n - 1] = (uint8_t)(out[n - 1] << (2 * l));
    }
    return n;
}

/* compress.rs:18-28: [node(kmer[..len-1]) || node(kmer[1..])] */
size_t ko_compress_kmer(const uint8_t *kmer, size_t len, uint8_t *out)
{
    if (len <= 2) { fprintf(stderr, "assertion failed: kmer.len() > 2\n"); abort(); }
    size_t n = ko_compress_node(kmer, len - 1, out);
    n += ko_compress_node(kmer + 1, len - 1, out + n);
    return n;
}

/* compress.rs:426-442 */
void ko_shift_right_bit_array(uint8_t *buf, size_t n, size_t shift_val)
{
    const size_t bits_in_carrier = 8;
    shift_val %= bits_in_carrier;
    if (shift_val == 0) return;
    uint8_t old_tmp = 0, new_tmp;
    size_t shift_remainder = bits_in_carrier - shift_val;
    uint8_t mask = (uint8_t)((1u << shift_val) - 1);
    for (size_t i = 0; i < n; ++i) {
        new_tmp = buf[i] & mask;
        buf[i] = (uint8_t)(buf[i] >> shift_val);
        buf[i] |= (uint8_t)(old_tmp << shift_remainder);
        old_tmp = new_tmp;
    }
}

/* compress.rs:403-419 */
void ko_shift_left_bit_array(uint8_t *buf, size_t n, size_t shift_val)
{
    const size_t bits_in_carrier = 8;
    shift_val %= bits_in_carrier;
    if (shift_val == 0) return;
    uint8_t old_tmp = 0, new_tmp;
    size_t shift_remainder = bits_in_carrier - shift_val;
    uint8_t mask = (uint8_t)(((1u << shift_val) - 1) << shift_remainder);
    for (size_t i = n; i-- > 0;) {
        new_tmp = buf[i] & mask;
        buf[i] = (uint8_t)(buf[i] << shift_val);
        buf[i] |= (uint8_t)(old_tmp >> shift_remainder);
        old_tmp = new_tmp;
    }
}

/* compress.rs:121-131 (Reverse for u8): swap 2-bit groups inside the byte */
static uint8_t reverse_u8(uint8_t x)
{
    x = (uint8_t)(((x >> 2) & 0x33) | ((x & 0x33) << 2));
    x = (uint8_t)(((x >> 4) & 0x0F) | ((x & 0x0F) << 4));
    return x;
}

/* compress.rs:153-169 */
void ko_reverse_compressed_node(const uint8_t *compr, size_t n, size_t remainder_size, uint8_t *out)
{
    size_t padding = ((CHARS_PER_CARRIER - remainder_size) % CHARS_PER_CARRIER) * 2;
    memmove(out, compr, n);
    size_t last_byte = n - 1;
    ko_shift_right_bit_array(out, n, padding);
    for (size_t i = 0; i < n / 2; ++i) { /* reversed.reverse() */
        uint8_t t = out[i];
        out[i] = out[n - 1 - i];
        out[n - 1 - i] = t;
    }
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)~reverse_u8(out[i]);
    out[last_byte] &= (uint8_t)~((1u << padding) - 1);
}

/* compress.rs:34-48: rc k-mer = [rc(target node) || rc(source node)] */
size_t ko_compress_kmer_with_rev_compl(const uint8_t *kmer, size_t len, uint8_t *out, uint8_t *rev)
{
    size_t n = ko_compress_kmer(kmer, len, out);
    size_t remainder = K1_SIZE % CHARS_PER_CARRIER;
    ko_reverse_compressed_node(out + COMPRESSED_K1_SIZE, COMPRESSED_K1_SIZE, remainder, rev);
    ko_reverse_compressed_node(out, COMPRESSED_K1_SIZE, remainder, rev + COMPRESSED_K1_SIZE);
    return n;
}

/* compress.rs:381-397 */
void ko_decode_compressed_chunk(uint8_t chunk, uint8_t out[4])
{
    static const uint8_t sym[4] = {'A', 'C', 'G', 'T'};
    for (int i = CHARS_PER_CARRIER - 1; i >= 0; --i) {
        out[i] = sym[chunk & 3];
        chunk >>= 2;
    }
}

/* compress.rs:304-314 */
char ko_decompress_char(uint8_t chunk, size_t padding)
{
    static const char sym[4] = {'A', 'C', 'G', 'T'};
    chunk = (uint8_t)(chunk >> (2 * padding));
    return sym[chunk & 3];
}

/* compress.rs:78-85 */
size_t ko_decompress_node(const uint8_t *node, size_t n, uint8_t *out)
{
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        ko_decode_compressed_chunk(node[i], out + m);
        m += 4;
    }
    return m < K1_SIZE ? m : K1_SIZE; /* truncate(K1_SIZE) */
}

/* compress.rs:100-105 */
static uint8_t get_last_char_from_node(const uint8_t *node, size_t n)
{
    size_t padding = K1_SIZE % CHARS_PER_CARRIER;
    uint8_t last_carrier = node[n - 1];
    padding = (CHARS_PER_CARRIER - padding) % CHARS_PER_CARRIER;
    return (uint8_t)ko_decompress_char(last_carrier, padding);
}

/* compress.rs:90-97 */
size_t ko_decompress_kmer(const uint8_t *kmer, size_t n, uint8_t *out)
{
    uint8_t tmp[4 * 64];
    size_t m = ko_decompress_node(kmer, COMPRESSED_K1_SIZE, tmp);
    memcpy(out, tmp, m);
    out[m] = get_last_char_from_node(kmer + COMPRESSED_K1_SIZE, n - COMPRESSED_K1_SIZE);
    return m + 1;
}

/* compress.rs:250-271 */
size_t ko_compress_edge(const uint8_t *edge, size_t len, uint8_t *out)
{
    if (len == 0) { fprintf(stderr, "assertion failed: edge.len() > 0\n"); abort(); }
    size_t compressed_size = 1 + ceil_div(len, CHARS_PER_CARRIER);
    size_t n = 0;
    out[n++] = 0;
    for (size_t i = 0; i < len; i += CHARS_PER_CARRIER) {
        uint8_t byte = 0;
        size_t chunk = len - i < CHARS_PER_CARRIER ? len - i : CHARS_PER_CARRIER;
        for (size_t j = 0; j < chunk; ++j) byte = ko_encode_fasta_symbol(edge[i + j], byte);
        out[n++] = byte;
    }
    size_t last = len % CHARS_PER_CARRIER ? len % CHARS_PER_CARRIER : CHARS_PER_CARRIER;
    size_t padding = CHARS_PER_CARRIER - last;
    out[compressed_size - 1] = (uint8_t)(out[compressed_size - 1] << (2 * padding));
    out[0] = (uint8_t)padding;
    return compressed_size;
}

/* compress.rs:283-293 */
size_t ko_decompress_edge(const uint8_t *edge, size_t n, uint8_t *out)
{
    size_t padding = edge[0];
    size_t m = 0;
    for (size_t i = 1; i < n; ++i) {
        ko_decode_compressed_chunk(edge[i], out + m);
        m += 4;
    }
    return m - padding;
}

/* compress.rs:231-233 */
size_t ko_kmer_to_edge(const uint8_t *kmer, size_t n, uint8_t *out)
{
    uint8_t tmp[4 * 64 + 4];
    size_t m = ko_decompress_kmer(kmer, n, tmp);
    return ko_compress_edge(tmp, m, out);
}

/* compress.rs:109-115 */
uint8_t ko_change_char_in_chunk(uint8_t chunk, size_t offset, uint8_t to)
{
    uint8_t mask = (uint8_t)((0xFFu - 3u) << (2 * offset));
    uint8_t compressed_char = (uint8_t)(ko_encode_fasta_symbol(to, 0) << (2 * offset));
    /* Rust `(CDC::max_value() - 3) << (2*offset)` shifts zeros in from the right, so the
     * bits BELOW the symbol are cleared too (they are padding, i.e. zero, by contract). */
    chunk &= mask;
    chunk |= compressed_char;
    return chunk;
}

/* compress.rs:193-201 */
size_t ko_change_last_char_in_edge(const uint8_t *edge, size_t n, uint8_t to, uint8_t *out)
{
    memmove(out, edge, n);
    size_t padding = out[0];
    out[n - 1] = ko_change_char_in_chunk(out[n - 1], padding, to);
    return n;
}

/* compress.rs:205-227 */
size_t ko_add_char_to_edge(const uint8_t *edge, size_t n, uint8_t chr, uint8_t *out)
{
    if (n <= 1) { fprintf(stderr, "assertion failed: edge.len() > 1\n"); abort(); }
    uint8_t padding = edge[0];
    size_t len = n - 1;
    uint8_t new_pad = (uint8_t)((uint8_t)(padding - 1) % CHARS_PER_CARRIER);
    uint8_t mask = (uint8_t)(0xFCu << (2 * new_pad));
    chr = ko_encode_fasta_symbol(chr, 0);
    memmove(out, edge, n);
    if (new_pad != 3) {
        out[len] &= mask;
        out[len] |= (uint8_t)(chr << (2 * new_pad));
        out[0] = new_pad;
        return n;
    }
    out[0] = new_pad;
    out[n] = (uint8_t)(chr << (2 * new_pad));
    return n + 1;
}

/* compress.rs:174-189 */
size_t ko_extend_edge(const uint8_t *edge, size_t n, const uint8_t *with, size_t with_len, uint8_t *out)
{
    uint8_t padding = edge[0];
    size_t vn = n;
    uint8_t *rem = (uint8_t *)malloc(4 + with_len);
    size_t rn = 0;
    memmove(out, edge, n);
    if (padding != 0) {
        size_t keep = (CHARS_PER_CARRIER - (size_t)edge[0]) % CHARS_PER_CARRIER;
        uint8_t dec[4];
        ko_decode_compressed_chunk(out[--vn], dec); /* vec.pop() */
        memcpy(rem, dec, keep);
        rn = keep;
    }
    memcpy(rem + rn, with, with_len);
    rn += with_len;
    uint8_t *compressed = (uint8_t *)malloc(2 + rn / 4 + 1);
    size_t cn = ko_compress_edge(rem, rn, compressed);
    out[0] = compressed[0];
    memcpy(out + vn, compressed + 1, cn - 1);
    vn += cn - 1;
    free(rem);
    free(compressed);
    return vn;
}

/* ============================ small containers ==================================== */

#define END ((uint64_t)-1) /* petgraph EdgeIndex::end() */

static void *xrealloc(void *p, size_t n)
{
    void *q = realloc(p, n ? n : 1);
    if (!q) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return q;
}

static uint64_t fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}

/* SEQUENCES (asm/mod.rs:15-26, prelude.rs:26-29): Vec<Box<[u8]>>; during the build every
 * slot holds one k-mer in compress_kmer format, so a flat array of fixed-size slots is
 * the same container.  Slot 0 is the scratch slot.                                      */
typedef struct {
    uint8_t *data;
    size_t len, cap, slot;
} seqs_t;

static void seqs_init(seqs_t *s, size_t slot)
{
    s->slot = slot; s->len = 1; s->cap = 1024;
    s->data = (uint8_t *)xrealloc(NULL, s->cap * slot);
    memset(s->data, 0, slot);
}
static size_t seqs_push(seqs_t *s, const uint8_t *bytes)
{
    if (s->len == s->cap) {
        /* `bytes` may point into s->data (the GIR pushes a clone of scratch slot 0) */
        size_t inside = (bytes >= s->data && bytes < s->data + s->len * s->slot) ? (size_t)(bytes - s->data) + 1 : 0;
        s->cap *= 2; s->data = (uint8_t *)xrealloc(s->data, s->cap * s->slot);
        if (inside) bytes = s->data + (inside - 1);
    }
    memmove(s->data + s->len * s->slot, bytes, s->slot);
    return s->len++;
}
/* NodeSlice bytes: slices.rs:138-145 (idx = offset/2, half = offset%2) */
static const uint8_t *node_bytes(const seqs_t *s, uint64_t offset)
{
    return s->data + (offset / 2) * s->slot + (offset % 2) * COMPRESSED_K1_SIZE;
}

/* HashMap<NodeSlice, V>: keys are NodeSlice offsets, hashed/compared through SEQUENCES
 * (slices.rs:94-108).  Open addressing; value is a u64.                                */
typedef struct {
    uint64_t *key; /* offset + 1, 0 = empty */
    uint64_t *val;
    size_t cap, len;
} nmap_t;

static void nmap_init(nmap_t *m) { m->cap = 1024; m->len = 0; m->key = (uint64_t *)calloc(m->cap, 8); m->val = (uint64_t *)calloc(m->cap, 8); }
static void nmap_free(nmap_t *m) { free(m->key); free(m->val); }
static size_t nmap_find_slot(const nmap_t *m, const seqs_t *s, const uint8_t *bytes, int *found)
{
    size_t i = fnv1a(bytes, COMPRESSED_K1_SIZE) & (m->cap - 1);
    for (;;) {
        if (!m->key[i]) { *found = 0; return i; }
        if (!memcmp(node_bytes(s, m->key[i] - 1), bytes, COMPRESSED_K1_SIZE)) { *found = 1; return i; }
        i = (i + 1) & (m->cap - 1);
    }
}
static void nmap_grow(nmap_t *m, const seqs_t *s)
{
    nmap_t n; n.cap = m->cap * 2; n.len = m->len;
    n.key = (uint64_t *)calloc(n.cap, 8); n.val = (uint64_t *)calloc(n.cap, 8);
    if (!n.key || !n.val) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    for (size_t i = 0; i < m->cap; ++i) if (m->key[i]) {
        int f; size_t j = nmap_find_slot(&n, s, node_bytes(s, m->key[i] - 1), &f);
        n.key[j] = m->key[i]; n.val[j] = m->val[i];
    }
    nmap_free(m); *m = n;
}

/* ====================== petgraph::Graph restatement (0.4.13) ======================= */
typedef struct {
    uint64_t n_nodes, cap_nodes;
    uint64_t *node_next[2];          /* head of outgoing / incoming edge lists */
    uint64_t n_edges, cap_edges;
    uint64_t *edge_next[2], *edge_node[2];
    uint32_t *edge_w;
    uint64_t *edge_slot;             /* EdgeSlice offset */
} pgraph_t;

static void pg_init(pgraph_t *g) { memset(g, 0, sizeof *g); }
static void pg_free(pgraph_t *g)
{
    for (int d = 0; d < 2; ++d) { free(g->node_next[d]); free(g->edge_next[d]); free(g->edge_node[d]); }
    free(g->edge_w); free(g->edge_slot);
}
static uint64_t pg_add_node(pgraph_t *g)
{
    if (g->n_nodes == g->cap_nodes) {
        g->cap_nodes = g->cap_nodes ? g->cap_nodes * 2 : 1024;
        for (int d = 0; d < 2; ++d) g->node_next[d] = (uint64_t *)xrealloc(g->node_next[d], g->cap_nodes * 8);
    }
    g->node_next[0][g->n_nodes] = END; g->node_next[1][g->n_nodes] = END;
    return g->n_nodes++;
}
/* petgraph Graph::add_edge: the new edge is linked at the HEAD of a's outgoing list and
 * of b's incoming list (so first_edge() returns the most recently added edge).          */
static uint64_t pg_add_edge(pgraph_t *g, uint64_t a, uint64_t b, uint64_t slot, uint32_t w)
{
    if (g->n_edges == g->cap_edges) {
        g->cap_edges = g->cap_edges ? g->cap_edges * 2 : 1024;
        for (int d = 0; d < 2; ++d) {
            g->edge_next[d] = (uint64_t *)xrealloc(g->edge_next[d], g->cap_edges * 8);
            g->edge_node[d] = (uint64_t *)xrealloc(g->edge_node[d], g->cap_edges * 8);
        }
        g->edge_w = (uint32_t *)xrealloc(g->edge_w, g->cap_edges * 4);
        g->edge_slot = (uint64_t *)xrealloc(g->edge_slot, g->cap_edges * 8);
    }
    uint64_t e = g->n_edges++;
    g->edge_node[0][e] = a; g->edge_node[1][e] = b;
    g->edge_w[e] = w; g->edge_slot[e] = slot;
    g->edge_next[0][e] = g->node_next[0][a];
    g->edge_next[1][e] = g->node_next[1][b];
    g->node_next[0][a] = e;
    g->node_next[1][b] = e;
    return e;
}
/* petgraph Graph::find_edge (directed): walk a's outgoing list for target b */
static uint64_t pg_find_edge(const pgraph_t *g, uint64_t a, uint64_t b)
{
    for (uint64_t e = g->node_next[0][a]; e != END; e = g->edge_next[0][e])
        if (g->edge_node[1][e] == b) return e;
    return END;
}
static uint64_t pg_degree(const pgraph_t *g, uint64_t n, int dir)
{
    uint64_t c = 0;
    for (uint64_t e = g->node_next[dir][n]; e != END; e = g->edge_next[dir][e]) ++c;
    return c;
}

/* ---- petgraph 0.4.13 removal, restated from its published source (crate absent): edges and nodes live in Vecs,
 * removal is swap_remove (the last element takes the freed index), adjacency is two singly linked lists per node. */
static void pg_change_edge_links(pgraph_t *g, const uint64_t edge_node[2], uint64_t e, const uint64_t edge_next[2])
{
    for (int k = 0; k < 2; ++k) {
        if (edge_node[k] >= g->n_nodes) return;
        uint64_t fst = g->node_next[k][edge_node[k]];
        if (fst == e) {
            g->node_next[k][edge_node[k]] = edge_next[k];
        } else {
            for (uint64_t cur = fst; cur != END && cur < g->n_edges; cur = g->edge_next[k][cur])
                if (g->edge_next[k][cur] == e) { g->edge_next[k][cur] = edge_next[k]; break; }
        }
    }
}
/* Graph::remove_edge + remove_edge_adjust_indices; returns 0 if e is out of range */
static int pg_remove_edge(pgraph_t *g, uint64_t e)
{
    if (e >= g->n_edges) return 0;
    uint64_t en[2] = {g->edge_node[0][e], g->edge_node[1][e]}, nx[2] = {g->edge_next[0][e], g->edge_next[1][e]};
    pg_change_edge_links(g, en, e, nx);
    uint64_t last = g->n_edges - 1;                          /* swap_remove */
    if (e != last) {
        for (int k = 0; k < 2; ++k) { g->edge_node[k][e] = g->edge_node[k][last]; g->edge_next[k][e] = g->edge_next[k][last]; }
        g->edge_w[e] = g->edge_w[last]; g->edge_slot[e] = g->edge_slot[last];
    }
    g->n_edges = last;
    if (e < g->n_edges) {                                    /* an edge was swapped in: relink `last` -> `e` */
        uint64_t swap[2] = {g->edge_node[0][e], g->edge_node[1][e]}, ee[2] = {e, e};
        pg_change_edge_links(g, swap, last, ee);
    }
    return 1;
}
static void pg_remove_node(pgraph_t *g, uint64_t a)
{
    if (a >= g->n_nodes) return;
    for (int k = 0; k < 2; ++k)
        while (g->node_next[k][a] != END) pg_remove_edge(g, g->node_next[k][a]);
    uint64_t last = g->n_nodes - 1;                          /* swap_remove */
    if (a != last) { g->node_next[0][a] = g->node_next[0][last]; g->node_next[1][a] = g->node_next[1][last]; }
    g->n_nodes = last;
    if (a < g->n_nodes)                                      /* the relocated node's edges point at its new index */
        for (int k = 0; k < 2; ++k)
            for (uint64_t cur = g->node_next[k][a]; cur != END; cur = g->edge_next[k][cur]) g->edge_node[k][cur] = a;
}
static int pg_degree_at_least(const pgraph_t *g, uint64_t n, int dir, uint64_t want)
{
    if (n >= g->n_nodes) return want == 0;
    uint64_t c = 0;
    for (uint64_t e = g->node_next[dir][n]; e != END; e = g->edge_next[dir][e]) if (++c >= want) return 1;
    return c >= want;
}

/* ============================ pruner.rs:36-82,199-257 (PtGraph) ===================== */
typedef struct { uint64_t *v; size_t n, cap; } evec_t;
static void evec_push(evec_t *v, uint64_t x)
{
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 256; v->v = (uint64_t *)xrealloc(v->v, v->cap * 8); }
    v->v[v->n++] = x;
}
/* pruner.rs:229-257; dir0 = first_direction, dir1 = second_direction (0 Outgoing, 1 Incoming) */
static void check_dead_path(const pgraph_t *g, uint64_t vertex, int first_direction, int second_direction, evec_t *out)
{
    uint64_t current = vertex, cnt = 0;
    for (;;) {
        cnt += 1;
        if (cnt >= 2 * K_SIZE) { out->n = 0; return; }           /* this path is not dead */
        uint64_t e = g->node_next[second_direction][current];     /* first_edge(current, second_direction) */
        if (e != END) {
            evec_push(out, e);
            current = g->edge_node[1][e];                         /* edge_endpoints(e).1 */
        } else {
            return;
        }
        if (pg_degree_at_least(g, current, first_direction, 3)) return;   /* neighbors_directed(..).nth(2).is_some() */
    }
}
static int cmp_desc(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? 1 : x > y ? -1 : 0;
}
static void remove_single_node(pgraph_t *g, uint64_t node)      /* pruner.rs:219-225 */
{
    if (!pg_degree_at_least(g, node, 1, 1) && !pg_degree_at_least(g, node, 0, 1)) pg_remove_node(g, node);
}
/* Prunable::remove_dead_paths for PtGraph (pruner.rs:36-82) with remove_paths (199-217) */
static uint64_t g_prune_passes = 0;      /* iterations of the outer loop in the last remove_dead_paths (the empty one included) */
uint64_t ko_last_prune_passes(void) { return g_prune_passes; }
static void remove_dead_paths(pgraph_t *g)
{
    evec_t to_remove = {0, 0, 0}, path = {0, 0, 0};
    g_prune_passes = 0;
    for (;;) {
        g_prune_passes += 1;
        for (uint64_t v = 0; v < g->n_nodes; ++v) {               /* Externals (pruner.rs:165-195) */
            path.n = 0;
            if (g->node_next[1][v] == END) check_dead_path(g, v, 1, 0, &path);          /* Input: no incoming edge */
            else if (g->node_next[0][v] == END) check_dead_path(g, v, 0, 1, &path);     /* Output: no outgoing edge */
            else continue;
            for (size_t i = 0; i < path.n; ++i) evec_push(&to_remove, path.v[i]);
        }
        if (to_remove.n == 0) break;
        qsort(to_remove.v, to_remove.n, 8, cmp_desc);
        for (size_t i = 0; i < to_remove.n; ++i) {                /* remove_paths */
            uint64_t e = to_remove.v[i];
            int have = e < g->n_edges;
            uint64_t a = have ? g->edge_node[0][e] : 0, b = have ? g->edge_node[1][e] : 0;
            pg_remove_edge(g, e);
            if (have) {
                if (a < b) { remove_single_node(g, b); remove_single_node(g, a); }
                else { remove_single_node(g, a); remove_single_node(g, b); }
            }
        }
        to_remove.n = 0;
    }
    free(to_remove.v); free(path.v);
}

/* ==================== PtGraphBuilder (pt_graph.rs:104-110) ========================= */
typedef struct {
    pgraph_t graph;
    nmap_t reads_to_nodes;   /* pt_graph.rs:87 */
    seqs_t seqs;             /* the global SEQUENCES */
} builder_t;

/* pt_graph.rs:142-154 */
static uint64_t add_fasta_node(builder_t *b, uint64_t node_offset)
{
    int found;
    size_t i = nmap_find_slot(&b->reads_to_nodes, &b->seqs, node_bytes(&b->seqs, node_offset), &found);
    if (found) return b->reads_to_nodes.val[i];
    uint64_t idx = pg_add_node(&b->graph);
    b->reads_to_nodes.key[i] = node_offset + 1;
    b->reads_to_nodes.val[i] = idx;
    if (++b->reads_to_nodes.len * 2 > b->reads_to_nodes.cap) nmap_grow(&b->reads_to_nodes, &b->seqs);
    return idx;
}

/* pt_graph.rs:172-198 */
static void add_single_edge_fastaq(builder_t *b, int first_edge, const uint8_t *compressed,
                                   uint64_t *s, uint64_t *t)
{
    uint64_t offset = seqs_push(&b->seqs, compressed);
    if (first_edge) *s = add_fasta_node(b, 2 * offset);
    *t = add_fasta_node(b, 2 * offset + 1);
    uint64_t e = pg_find_edge(&b->graph, *s, *t);
    if (e != END) {
        b->seqs.len--;              /* SEQUENCES.write().pop() */
        b->graph.edge_w[e] += 1;    /* u32 `+= 1` (wraps in release builds) */
    } else {
        pg_add_edge(&b->graph, *s, *t, offset, 1);
    }
    *s = *t;
}

/* pt_graph.rs:277-315 */
static int add_read_fastaq(builder_t *b, const uint8_t *read, size_t len, int reverse_complement)
{
    if (len < K_SIZE) { snprintf(g_err, sizeof g_err, "Read is too short!"); return KO_E_SHORT_READ; }
    uint64_t s = 0, t = 0;
    size_t slot = b->seqs.slot, n_win = len - K_SIZE + 1;
    uint8_t kbuf[64];
    if (reverse_complement) {
        uint8_t *reversed = (uint8_t *)xrealloc(NULL, n_win * slot);
        reversed[(n_win - 1) * slot] = 0;   /* (n_win >= 1: tells the compiler the last slot is written before it is read) */
        for (size_t cnt = 0; cnt < n_win; ++cnt) {
            ko_compress_kmer_with_rev_compl(read + cnt, K_SIZE, kbuf, reversed + cnt * slot);
            add_single_edge_fastaq(b, cnt == 0, kbuf, &s, &t);
        }
        /* reversed.remove(last) first, then drain(..).rev(): i.e. last window first */
        size_t rev = n_win - 1;
        add_single_edge_fastaq(b, 1, reversed + rev * slot, &s, &t);
        for (size_t i = rev; i-- > 0;) add_single_edge_fastaq(b, 0, reversed + i * slot, &s, &t);
        free(reversed);
    } else {
        for (size_t cnt = 0; cnt < n_win; ++cnt) {
            ko_compress_kmer(read + cnt, K_SIZE, kbuf);
            add_single_edge_fastaq(b, cnt == 0, kbuf, &s, &t);
        }
    }
    return KO_OK;
}

/* pt_graph.rs:201-213 add_single_edge_bfc: push the slot, find-or-add both nodes (add_bfc_node 119-139: the
 * fixedbitset/region arithmetic of get_node_idx 157-169 is just "index in first-seen order", which the map
 * gives directly), and ALWAYS add an edge -- no find_edge, duplicates become parallel edges.              */
static void add_single_edge_bfc(builder_t *b, const uint8_t *compressed, uint32_t weight)
{
    uint64_t offset = seqs_push(&b->seqs, compressed);
    uint64_t s = add_fasta_node(b, 2 * offset);
    uint64_t t = add_fasta_node(b, 2 * offset + 1);
    pg_add_edge(&b->graph, s, t, offset, weight);
}
/* pt_graph.rs:317-330 */
static int add_read_bfc(builder_t *b, const uint8_t *read, size_t len, uint32_t weight, int reverse_complement)
{
    if (len < K_SIZE) { snprintf(g_err, sizeof g_err, "Read is too short!"); return KO_E_SHORT_READ; }
    if (len != K_SIZE) { snprintf(g_err, sizeof g_err, "oracle: BFCounter line of %zu bases with k=%zu is not restated", len, K_SIZE); return KO_E_ARG; }
    uint8_t kbuf[64], rbuf[64];
    if (reverse_complement) {
        ko_compress_kmer_with_rev_compl(read, len, kbuf, rbuf);
        add_single_edge_bfc(b, kbuf, weight);
        add_single_edge_bfc(b, rbuf, weight);
    } else {
        ko_compress_kmer(read, len, kbuf);
        add_single_edge_bfc(b, kbuf, weight);
    }
    return KO_OK;
}

/* ============ HmGIR (hm_gir.rs:22,39-153; hs_gir.rs:192-203) -- counts only ========= */
typedef struct { uint64_t to; uint32_t w; uint8_t last_char; } gedge_t;
typedef struct { gedge_t *e; uint32_t n; } outgoing_t;
typedef struct {
    nmap_t map;          /* NodeSlice -> index into out[] */
    outgoing_t *out; size_t n_out, cap_out;
    seqs_t seqs;
} gir_t;

static uint64_t gir_insert(gir_t *g, uint64_t node_offset)
{
    int found;
    size_t i = nmap_find_slot(&g->map, &g->seqs, node_bytes(&g->seqs, node_offset), &found);
    if (g->n_out == g->cap_out) { g->cap_out = g->cap_out ? g->cap_out * 2 : 1024; g->out = (outgoing_t *)xrealloc(g->out, g->cap_out * sizeof(outgoing_t)); }
    g->out[g->n_out].e = NULL; g->out[g->n_out].n = 0;
    g->map.key[i] = node_offset + 1; g->map.val[i] = g->n_out++;
    if (++g->map.len * 2 > g->map.cap) nmap_grow(&g->map, &g->seqs);
    (void)found;
    return node_offset;
}
/* hs_gir.rs:192-203 */
static void create_or_modify_edge(outgoing_t *o, uint64_t to, uint8_t last_char)
{
    for (uint32_t i = 0; i < o->n; ++i) if (o->e[i].to == to) { o->e[i].w += 1; return; }
    o->e = (gedge_t *)xrealloc(o->e, (o->n + 1) * sizeof(gedge_t));
    o->e[o->n].to = to; o->e[o->n].w = 1; o->e[o->n].last_char = last_char; o->n++;
}
/* hm_gir.rs:91-153 */
static void gir_add_single_edge(gir_t *g, int first_node, const uint8_t *compressed,
                                uint64_t *source_node, uint64_t *target_node, uint8_t last_char)
{
    int insert = 0, found;
    memcpy(g->seqs.data, compressed, g->seqs.slot);          /* s[0] = compressed */
    if (first_node) {
        size_t i = nmap_find_slot(&g->map, &g->seqs, node_bytes(&g->seqs, 0), &found);
        if (found) *source_node = g->map.key[i] - 1;
        else {
            uint64_t offset = seqs_push(&g->seqs, g->seqs.data);
            insert = 1;
            *source_node = 2 * offset;
        }
        if (insert) gir_insert(g, *source_node);
    }
    {
        size_t i = nmap_find_slot(&g->map, &g->seqs, node_bytes(&g->seqs, 1), &found);
        if (found) { insert = 0; *target_node = g->map.key[i] - 1; }
        else {
            uint64_t offset;
            if (!insert) { seqs_push(&g->seqs, g->seqs.data); offset = 2 * g->seqs.len - 1; }
            else offset = *source_node + 1;
            insert = 1;
            *target_node = offset;
        }
    }
    if (insert) gir_insert(g, *target_node);
    int f; size_t si = nmap_find_slot(&g->map, &g->seqs, node_bytes(&g->seqs, *source_node), &f);
    if (!f) { fprintf(stderr, "Node disappeared\n"); abort(); }
    create_or_modify_edge(&g->out[g->map.val[si]], *target_node, last_char);
    *source_node = *target_node;
}
/* hm_gir.rs:39-87 */
static int gir_add_read_fastaq(gir_t *g, const uint8_t *read, size_t len, int reverse_complement)
{
    if (len < K_SIZE) { snprintf(g_err, sizeof g_err, "Read is too short!"); return KO_E_SHORT_READ; }
    uint64_t s = 0, t = 0;
    size_t slot = g->seqs.slot, n_win = len - K_SIZE + 1;
    uint8_t kbuf[64];
    if (reverse_complement) {
        uint8_t *reversed = (uint8_t *)xrealloc(NULL, n_win * slot);
        reversed[(n_win - 1) * slot] = 0;   /* (n_win >= 1: tells the compiler the last slot is written before it is read) */
        for (size_t cnt = 0; cnt < n_win; ++cnt) {
            ko_compress_kmer_with_rev_compl(read + cnt, K_SIZE, kbuf, reversed + cnt * slot);
            gir_add_single_edge(g, cnt == 0, kbuf, &s, &t, read[cnt + K_SIZE - 1]);
        }
        size_t rev = n_win - 1;
        gir_add_single_edge(g, 1, reversed + rev * slot, &s, &t, read[rev + K_SIZE - 1]);
        for (size_t i = rev; i-- > 0;) gir_add_single_edge(g, 0, reversed + i * slot, &s, &t, read[i + K_SIZE - 1]);
        free(reversed);
    } else {
        for (size_t cnt = 0; cnt < n_win; ++cnt) {
            ko_compress_kmer(read + cnt, K_SIZE, kbuf);
            gir_add_single_edge(g, cnt == 0, kbuf, &s, &t, read[cnt + K_SIZE - 1]);
        }
    }
    return KO_OK;
}

/* ======================= file ingest (builder.rs:42-77,118-165) ==================== */

/* builder.rs:57-77 */
static int check_file(const char *path, char *resolved)
{
    if (!realpath(path, resolved)) { snprintf(g_err, sizeof g_err, "Coulndt resolve path: %s", path); return KO_E_PATH; }
    struct stat st;
    if (stat(resolved, &st) != 0) { snprintf(g_err, sizeof g_err, "%s does not exist", resolved); return KO_E_NOT_EXIST; }
    if (S_ISDIR(st.st_mode)) { snprintf(g_err, sizeof g_err, "%s is a directory", resolved); return KO_E_IS_DIR; }
    return KO_OK;
}

typedef struct { uint8_t *buf; size_t len, pos; } linebuf_t;

static int slurp(const char *path, linebuf_t *lb)
{
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(g_err, sizeof g_err, "Couldn't open all files: %s", path); return KO_E_OPEN; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    lb->buf = (uint8_t *)xrealloc(NULL, (size_t)n + 1);
    lb->len = fread(lb->buf, 1, (size_t)n, f); lb->pos = 0;
    fclose(f);
    return KO_OK;
}
/* BufRead::read_line: up to and including '\n'; empty at EOF */
static void read_line(linebuf_t *lb, const uint8_t **p, size_t *n)
{
    *p = lb->buf + lb->pos;
    size_t i = lb->pos;
    while (i < lb->len && lb->buf[i] != '\n') ++i;
    if (i < lb->len) ++i;
    *n = i - lb->pos;
    lb->pos = i;
}
/* str::trim_right(): strips trailing Unicode whitespace; for ASCII input that is
 * space, \t, \n, \v, \f, \r                                                          */
static size_t trim_right(const uint8_t *p, size_t n)
{
    while (n && (p[n - 1] == ' ' || (p[n - 1] >= 9 && p[n - 1] <= 13))) --n;
    return n;
}
/* builder.rs:131,155: seq.iter().all(|x| "ACGT".bytes().any(|i| i == x)) */
static int all_acgt(const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; ++i) if (p[i] != 'A' && p[i] != 'C' && p[i] != 'G' && p[i] != 'T') return 0;
    return 1;
}

typedef int (*read_cb)(void *ctx, const uint8_t *seq, size_t len);

/* bio 0.10.0 io::fastq::Reader::read + Records::next, restated from its published
 * behaviour (crate absent): header line must start with '@'; then seq line, separator
 * line, quality line; a record whose quality line is missing is an error; `seq()` is the
 * sequence line right-trimmed.  Iteration ends when the header line is empty (EOF).     */
static int scan_fastq(linebuf_t *lb, read_cb cb, void *ctx, uint64_t *n_records)
{
    for (;;) {
        const uint8_t *h, *s, *sep, *q; size_t hn, sn, sepn, qn;
        read_line(lb, &h, &hn);
        if (hn == 0) return KO_OK;
        if (h[0] != '@') { snprintf(g_err, sizeof g_err, "Expected @ at record start."); return KO_E_PARSE; }
        read_line(lb, &s, &sn);
        read_line(lb, &sep, &sepn);
        read_line(lb, &q, &qn);
        if (qn == 0) { snprintf(g_err, sizeof g_err, "Incomplete record."); return KO_E_PARSE; }
        ++*n_records;
        int rc = cb(ctx, s, trim_right(s, sn));
        if (rc) return rc;
    }
}
/* bio 0.10.0 io::fasta::Reader::read: header starts with '>'; sequence = the following
 * lines up to the next '>' (or EOF), each right-trimmed and concatenated.               */
static int scan_fasta(linebuf_t *lb, read_cb cb, void *ctx, uint64_t *n_records)
{
    const uint8_t *line; size_t ln;
    read_line(lb, &line, &ln);
    uint8_t *seq = NULL; size_t cap = 0;
    int rc = KO_OK;
    while (ln) {
        if (line[0] != '>') { snprintf(g_err, sizeof g_err, "Expected > at record start."); rc = KO_E_PARSE; break; }
        size_t n = 0;
        for (;;) {
            read_line(lb, &line, &ln);
            if (ln == 0 || line[0] == '>') break;
            size_t t = trim_right(line, ln);
            if (n + t > cap) { cap = (n + t) * 2 + 64; seq = (uint8_t *)xrealloc(seq, cap); }
            memcpy(seq + n, line, t); n += t;
        }
        ++*n_records;
        rc = cb(ctx, seq, n);
        if (rc) break;
    }
    free(seq);
    return rc;
}

typedef int (*bfc_cb)(void *ctx, const uint8_t *kmer, size_t len, uint32_t weight);
/* builder.rs:79-115 create_bfc: io::Lines (strips "\n" / "\r\n"), split('\t'), u32 parse, threshold */
static int scan_bfc(linebuf_t *lb, uint32_t threshold, bfc_cb cb, void *ctx, uint64_t *total)
{
    for (;;) {
        const uint8_t *line; size_t ln;
        read_line(lb, &line, &ln);
        if (ln == 0) return KO_OK;
        if (line[ln - 1] == '\n') --ln;
        if (ln && line[ln - 1] == '\r') --ln;
        const uint8_t *tab = (const uint8_t *)memchr(line, '\t', ln);
        if (!tab) { snprintf(g_err, sizeof g_err, "called `Option::unwrap()` on a `None` value"); return KO_E_PARSE; }
        size_t klen = (size_t)(tab - line), wn = ln - klen - 1;
        const uint8_t *w = tab + 1;
        const uint8_t *tab2 = (const uint8_t *)memchr(w, '\t', wn);
        if (tab2) wn = (size_t)(tab2 - w);
        uint64_t weight = 0; size_t i = 0;
        if (i < wn && w[i] == '+') ++i;
        if (i == wn) { snprintf(g_err, sizeof g_err, "Parse int error"); return KO_E_PARSE; }
        for (; i < wn; ++i) {
            if (w[i] < '0' || w[i] > '9') { snprintf(g_err, sizeof g_err, "Parse int error"); return KO_E_PARSE; }
            weight = weight * 10 + (uint64_t)(w[i] - '0');
            if (weight > 0xFFFFFFFFull) { snprintf(g_err, sizeof g_err, "Parse int error"); return KO_E_PARSE; }
        }
        if (weight < threshold) continue;
        *total += klen;
        int rc = cb(ctx, line, klen, (uint32_t)weight);
        if (rc) return rc;
    }
}

static int scan_paths(const char *const *paths, size_t n_paths, int file_type, read_cb cb, void *ctx, uint64_t *n_records)
{
    /* builder.rs:46 check_files runs over ALL inputs before any is opened */
    char (*resolved)[PATH_MAX] = (char (*)[PATH_MAX])xrealloc(NULL, (n_paths ? n_paths : 1) * PATH_MAX);
    int rc = KO_OK;
    for (size_t i = 0; i < n_paths && !rc; ++i) rc = check_file(paths[i], resolved[i]);
    if (file_type != 0 && file_type != 1 && !rc) { snprintf(g_err, sizeof g_err, "oracle: file type %d not restated", file_type); rc = KO_E_ARG; }
    for (size_t i = 0; i < n_paths && !rc; ++i) {
        linebuf_t lb;
        rc = slurp(resolved[i], &lb);
        if (rc) break;
        rc = file_type == 1 ? scan_fastq(&lb, cb, ctx, n_records) : scan_fasta(&lb, cb, ctx, n_records);
        free(lb.buf);
    }
    free(resolved);
    return rc;
}

/* ============================ build driver ========================================= */
typedef struct {
    builder_t b;
    gir_t gir;
    int with_gir, rc;
    uint64_t total;
} build_ctx;

/* builder.rs:152-160 loop body */
static int build_read_cb(void *vctx, const uint8_t *seq, size_t len)
{
    build_ctx *c = (build_ctx *)vctx;
    if (!all_acgt(seq, len)) return KO_OK;       /* `continue` */
    c->total += len;
    int rc = add_read_fastaq(&c->b, seq, len, c->rc);
    if (rc) return rc;
    if (c->with_gir) rc = gir_add_read_fastaq(&c->gir, seq, len, c->rc);
    return rc;
}

static void ctx_init(build_ctx *c, int rc, int with_gir)
{
    memset(c, 0, sizeof *c);
    c->rc = rc; c->with_gir = with_gir;
    pg_init(&c->b.graph); nmap_init(&c->b.reads_to_nodes); seqs_init(&c->b.seqs, 2 * COMPRESSED_K1_SIZE);
    if (with_gir) { nmap_init(&c->gir.map); seqs_init(&c->gir.seqs, 2 * COMPRESSED_K1_SIZE); }
}
static void ctx_free(build_ctx *c)
{
    pg_free(&c->b.graph); nmap_free(&c->b.reads_to_nodes); free(c->b.seqs.data);
    if (c->with_gir) {
        nmap_free(&c->gir.map); free(c->gir.seqs.data);
        for (size_t i = 0; i < c->gir.n_out; ++i) free(c->gir.out[i].e);
        free(c->gir.out);
    }
}

/* stats/collections.rs:137-168 */
static void pt_stats(const pgraph_t *g, ko_stats *st)
{
    memset(st, 0, sizeof *st);
    st->node_count = g->n_nodes; st->edge_count = g->n_edges;
    uint64_t sum_w = 0;
    for (uint64_t e = 0; e < g->n_edges; ++e) {
        if (g->edge_w[e] > st->max_edge_weight) st->max_edge_weight = g->edge_w[e];
        sum_w += g->edge_w[e];
    }
    st->avg_edge_weight = (double)sum_w / (double)g->n_edges;
    uint64_t sum_out = 0;
    for (uint64_t n = 0; n < g->n_nodes; ++n) {
        uint64_t od = pg_degree(g, n, 0), id = pg_degree(g, n, 1);
        if (od > st->max_out_degree) st->max_out_degree = od;
        if (id > st->max_in_degree) st->max_in_degree = id;
        sum_out += od;
        if (g->node_next[1][n] == END) st->incoming_vert_count++;   /* externals(Incoming) */
        if (g->node_next[0][n] == END) st->outgoing_vert_count++;   /* externals(Outgoing) */
    }
    st->avg_out_degree = (double)sum_out / (double)g->n_nodes;
}

/* PtGraph::create tail (pt_graph.rs:339-344): recode every slot kmer->edge format */
/* Clean for PtGraph (pruner.rs:84-93) over petgraph 0.4.13's retain_edges / retain_nodes, which visit the indices in
 * DESCENDING order and remove_edge / remove_node (swap_remove) the ones the predicate rejects */
static void remove_single_vertices(pgraph_t *g)                     /* pruner.rs:85-87 */
{
    for (uint64_t n = g->n_nodes; n-- > 0;)
        if (g->node_next[0][n] == END && g->node_next[1][n] == END) pg_remove_node(g, n);   /* neighbors_undirected(n).next().is_none() */
}
static void remove_weak_edges(pgraph_t *g, uint32_t threshold)      /* pruner.rs:89-92 */
{
    for (uint64_t e = g->n_edges; e-- > 0;)
        if (!(g->edge_w[e] >= threshold)) pg_remove_edge(g, e);
    remove_single_vertices(g);
}

/* ============================ standardizer.rs:41-128 (PtGraph) ====================== */
static uint64_t pg_degree_exact(const pgraph_t *g, uint64_t n, int dir);
static void remove_weak_edges(pgraph_t *g, uint32_t threshold);
static int is_ambiguous(const pgraph_t *g, uint64_t n)               /* pt_graph.rs:54-62 */
{
    uint64_t in = pg_degree_exact(g, n, 1), out = pg_degree_exact(g, n, 0);
    return (in > 1 || out > 1) || (in == 0 && out >= 1);
}
/* Standardizable::standardize_contigs (standardizer.rs:72-82) with get_contigs_from_node (84-107) and
 * standardize_contig (109-122).  The reference walks a HashSet of ambiguous nodes; contigs never share an edge, so the
 * order does not matter. */
static void standardize_contigs(pgraph_t *g)
{
    uint8_t *amb = (uint8_t *)calloc(g->n_nodes ? g->n_nodes : 1, 1);
    for (uint64_t n = 0; n < g->n_nodes; ++n) amb[n] = (uint8_t)is_ambiguous(g, n);
    evec_t contig = {0, 0, 0};
    for (uint64_t start = 0; start < g->n_nodes; ++start) {
        if (!amb[start]) continue;
        for (uint64_t e0 = g->node_next[0][start]; e0 != END; e0 = g->edge_next[0][e0]) {     /* neighbors_directed(start, Outgoing) */
            uint64_t current_node = g->edge_node[1][e0];
            uint64_t current_edge = pg_find_edge(g, start, current_node);
            contig.n = 0;
            for (;;) {
                evec_push(&contig, current_edge);
                if (pg_degree_exact(g, current_node, 0) != 1 || amb[current_node]) break;
                current_edge = g->node_next[0][current_node];
                current_node = g->edge_node[1][current_edge];
            }
            uint64_t sum = 0;
            for (size_t i = 0; i < contig.n; ++i) sum += g->edge_w[contig.v[i]];
            const uint32_t w = (uint32_t)round((double)sum / (double)contig.n);
            for (size_t i = 0; i < contig.n; ++i) g->edge_w[contig.v[i]] = w;
        }
    }
    free(contig.v); free(amb);
}
/* Standardizable::standardize_edges (standardizer.rs:42-70) */
static void standardize_edges(pgraph_t *g, uint64_t original_genome_length, uint64_t k_size, uint32_t threshold)
{
    uint64_t s = 0, l = 0;
    for (uint64_t e = 0; e < g->n_edges; ++e) { s += g->edge_w[e]; if (g->edge_w[e] < threshold) l += g->edge_w[e]; }
    const double p = (double)(original_genome_length - k_size) / (double)(s - l);      /* calculate_standardization_ratio (124-128) */
    for (uint64_t e = 0; e < g->n_edges; ++e) {
        const double scaled = round((double)g->edge_w[e] * p);
        uint32_t nw = scaled >= 4294967295.0 ? 4294967295u : scaled > 0 ? (uint32_t)scaled : 0u;   /* `as EdgeWeight` saturates */
        g->edge_w[e] = (nw == 0 && g->edge_w[e] >= threshold) ? 1 : nw;
    }
    remove_weak_edges(g, 1);
}

/* ============================ shrinker.rs:38-209 (PtGraph) ========================== */
/* After PtGraph::create every SEQUENCES slot holds one edge in compress_edge format (pt_graph.rs:339-343); shrink
 * merges them (EdgeSlice::merge, slices.rs:23-34), so slots get their own growable byte strings here. */
typedef struct { uint8_t **bytes; size_t *len; size_t n; } labels_t;
static void labels_free(labels_t *l)
{
    if (!l->bytes) return;
    for (size_t i = 0; i < l->n; ++i) free(l->bytes[i]);
    free(l->bytes); free(l->len); l->bytes = NULL; l->len = NULL; l->n = 0;
}
/* EdgeSlice::merge (slices.rs:23-34): self <- extend_edge(self, decompress_edge(other)[K1_SIZE..]); other <- empty */
static void label_merge(labels_t *l, uint64_t self_idx, uint64_t other_idx)
{
    size_t on = l->len[other_idx];
    uint8_t *ascii = (uint8_t *)xrealloc(NULL, on * 4 + 8);
    size_t an = ko_decompress_edge(l->bytes[other_idx], on, ascii);
    if (!(an > K1_SIZE)) { fprintf(stderr, "oracle: assertion failed: other_uncompressed.len() > K1_SIZE\n"); abort(); }
    uint8_t *out = (uint8_t *)xrealloc(NULL, l->len[self_idx] + (an - K1_SIZE) / 4 + 8);
    size_t n = ko_extend_edge(l->bytes[self_idx], l->len[self_idx], ascii + K1_SIZE, an - K1_SIZE, out);
    free(ascii);
    free(l->bytes[other_idx]); l->bytes[other_idx] = NULL; l->len[other_idx] = 0;
    free(l->bytes[self_idx]); l->bytes[self_idx] = out; l->len[self_idx] = n;
}
static uint64_t pg_degree_exact(const pgraph_t *g, uint64_t n, int dir)
{
    uint64_t c = 0;
    for (uint64_t e = g->node_next[dir][n]; e != END; e = g->edge_next[dir][e]) ++c;
    return c;
}
/* ShrinkTraverse (shrinker.rs:38-147) */
typedef struct { uint8_t *fb; size_t n; uint64_t *stack; size_t sp, cap; size_t node_offset; } shrink_traverse;
static void st_push(shrink_traverse *t, uint64_t v)
{
    if (t->sp == t->cap) { t->cap = t->cap ? t->cap * 2 : 256; t->stack = (uint64_t *)xrealloc(t->stack, t->cap * 8); }
    t->stack[t->sp++] = v;
}
static void st_new(shrink_traverse *t, const pgraph_t *g)      /* shrinker.rs:48-60 */
{
    memset(t, 0, sizeof *t);
    for (uint64_t n = 0; n < g->n_nodes; ++n) if (g->node_next[1][n] == END) st_push(t, n);   /* graph.externals(Incoming) */
    t->n = g->n_nodes;
    t->fb = (uint8_t *)calloc(t->n ? t->n : 1, 1);
}
static uint64_t st_next(shrink_traverse *t, const pgraph_t *g) /* shrinker.rs:62-135 */
{
    int iter = 0;
    evec_t single_nodes = {0, 0, 0};
    for (;;) {
        while (t->sp) {
            uint64_t current_node = t->stack[t->sp - 1];
            for (;;) {
                int new_ancestor = 0;
                for (uint64_t e = g->node_next[0][current_node]; e != END; e = g->edge_next[0][e]) {
                    uint64_t n = g->edge_node[1][e];
                    if (t->fb[n]) continue;
                    t->fb[n] = 1;
                    if (current_node == n) continue;
                    else if (pg_degree_exact(g, n, 0) == 1 && pg_degree_exact(g, n, 1) == 1) { free(single_nodes.v); return e; }
                    else { st_push(t, n); current_node = n; new_ancestor = 1; break; }
                }
                if (!new_ancestor) { t->sp--; t->fb[current_node] = 1; break; }
            }
        }
        /* components with a cycle at their root: the next unvisited node -- self.fb.zeros().skip(self.node_offset).enumerate() */
        size_t zeros_seen = 0, i = 0;
        for (size_t n = 0; n < t->n; ++n) {
            if (t->fb[n]) continue;
            if (zeros_seen++ < t->node_offset) continue;
            if (g->node_next[1][n] == END && g->node_next[0][n] == END) {
                evec_push(&single_nodes, n);
            } else {
                st_push(t, n);
                t->node_offset += i;
                iter = 1;
                break;
            }
            ++i;
        }
        if (!iter) break;
        iter = 0;
        for (size_t j = 0; j < single_nodes.n; ++j) t->fb[single_nodes.v[j]] = 1;
        single_nodes.n = 0;
    }
    free(single_nodes.v);
    return END;
}
/* Shrinkable::shrink_single_path (shrinker.rs:178-209) */
static uint64_t shrink_single_path(pgraph_t *g, labels_t *l, uint64_t base_edge)
{
    uint64_t start_node = g->edge_node[0][base_edge], mid_node = g->edge_node[1][base_edge];
    for (;;) {
        uint64_t next_edge = g->node_next[0][mid_node];                  /* first_edge(mid_node, Outgoing) */
        uint64_t base_slot = g->edge_slot[base_edge]; uint32_t base_w = g->edge_w[base_edge];
        uint64_t target = g->edge_node[1][next_edge];
        uint64_t next_slot;
        if (base_edge < next_edge) {                                     /* higher index first */
            next_slot = g->edge_slot[next_edge];
            pg_remove_edge(g, next_edge);
            pg_remove_edge(g, base_edge);
        } else if (base_edge == next_edge) {
            return base_edge;
        } else {
            pg_remove_edge(g, base_edge);
            next_slot = g->edge_slot[next_edge];                         /* remove_edge(next_edge) returns the weight found AT that index now */
            pg_remove_edge(g, next_edge);
        }
        label_merge(l, base_slot, next_slot);
        base_edge = pg_add_edge(g, start_node, target, base_slot, base_w);
        mid_node = target;
        if (pg_degree_exact(g, mid_node, 1) != 1 || pg_degree_exact(g, mid_node, 0) != 1 || mid_node == start_node) return base_edge;
    }
}
/* Shrinkable::shrink (shrinker.rs:165-176) */
static void shrink(pgraph_t *g, labels_t *l)
{
    shrink_traverse t; st_new(&t, g);
    for (uint64_t base_edge; (base_edge = st_next(&t, g)) != END;) shrink_single_path(g, l, base_edge);
    remove_single_vertices(g);
    free(t.fb); free(t.stack);
}

/* ============================ collapser.rs:29-273 (PtGraph) ========================= */
/* petgraph 0.4.13 tarjan_scc, restated from its published source (crate absent): recursive visit in node_identifiers()
 * order, neighbours in adjacency-list order; each SCC is pushed when its root is finished, its nodes in pop order */
typedef struct { int64_t *index; uint64_t *lowlink; uint8_t *on_stack; uint64_t *stack; size_t sp; uint64_t counter;
                 uint64_t *scc_nodes; size_t scc_n; size_t *scc_start; size_t n_sccs; } tarjan_t;
static void scc_visit(const pgraph_t *g, tarjan_t *d, uint64_t v)
{
    if (d->index[v] >= 0) return;
    const uint64_t v_index = d->counter;
    d->index[v] = (int64_t)v_index; d->lowlink[v] = v_index; d->on_stack[v] = 1;
    d->stack[d->sp++] = v;
    d->counter += 1;
    for (uint64_t e = g->node_next[0][v]; e != END; e = g->edge_next[0][e]) {       /* g.neighbors(v) */
        const uint64_t w = g->edge_node[1][e];
        if (d->index[w] < 0) {
            scc_visit(g, d, w);
            if (d->lowlink[w] < d->lowlink[v]) d->lowlink[v] = d->lowlink[w];
        } else if (d->on_stack[w]) {
            if ((uint64_t)d->index[w] < d->lowlink[v]) d->lowlink[v] = (uint64_t)d->index[w];
        }
    }
    if (d->lowlink[v] == v_index) {
        d->scc_start[d->n_sccs++] = d->scc_n;
        for (;;) {
            const uint64_t w = d->stack[--d->sp];
            d->on_stack[w] = 0;
            d->scc_nodes[d->scc_n++] = w;
            if (w == v) break;
        }
    }
}
/* unwrap!(tarjan_scc(&self).iter().last())[0] (collapser.rs:63) */
static uint64_t last_scc_first_node(const pgraph_t *g)
{
    const size_t n = g->n_nodes;
    tarjan_t d; memset(&d, 0, sizeof d);
    d.index = (int64_t *)xrealloc(NULL, n * 8); d.lowlink = (uint64_t *)xrealloc(NULL, n * 8); d.on_stack = (uint8_t *)calloc(n, 1);
    d.stack = (uint64_t *)xrealloc(NULL, n * 8); d.scc_nodes = (uint64_t *)xrealloc(NULL, n * 8); d.scc_start = (size_t *)xrealloc(NULL, n * sizeof(size_t));
    for (size_t i = 0; i < n; ++i) d.index[i] = -1;
    for (uint64_t v = 0; v < n; ++v) scc_visit(g, &d, v);
    const uint64_t r = d.scc_nodes[d.scc_start[d.n_sccs - 1]];
    free(d.index); free(d.lowlink); free(d.on_stack); free(d.stack); free(d.scc_nodes); free(d.scc_start);
    return r;
}

typedef struct { char **v; size_t n, cap; } strvec_t;
static void strvec_push(strvec_t *s, const char *str, size_t len)
{
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 64; s->v = (char **)xrealloc(s->v, s->cap * sizeof(char *)); }
    s->v[s->n] = (char *)xrealloc(NULL, len + 1); memcpy(s->v[s->n], str, len); s->v[s->n][len] = 0; s->n++;
}
typedef struct { char *p; size_t n, cap; } str_t;
static void str_append(str_t *s, const uint8_t *b, size_t len)
{
    if (s->n + len + 1 > s->cap) { s->cap = (s->n + len + 1) * 2; s->p = (char *)xrealloc(s->p, s->cap); }
    memcpy(s->p + s->n, b, len); s->n += len; s->p[s->n] = 0;
}
/* EdgeSlice::name (decompress_edge of the slot) appended whole, or from K1_SIZE on (remainder, slices.rs:36-42) */
static void append_edge_name(str_t *contig, const labels_t *l, uint64_t slot, int remainder_only)
{
    uint8_t *ascii = (uint8_t *)xrealloc(NULL, l->len[slot] * 4 + 8);
    const size_t an = ko_decompress_edge(l->bytes[slot], l->len[slot], ascii);
    if (remainder_only) str_append(contig, ascii + K1_SIZE, an - K1_SIZE); else str_append(contig, ascii, an);
    free(ascii);
}
static uint64_t self_loop(const pgraph_t *g, uint64_t node)           /* collapser.rs:213-224 */
{
    if (pg_degree_exact(g, node, 1) > 2) return END;
    for (uint64_t e = g->node_next[0][node]; e != END; e = g->edge_next[0][e]) if (g->edge_node[1][e] == g->edge_node[0][e]) return e;
    return END;
}
static uint64_t simple_loop(const pgraph_t *g, uint64_t edge)          /* collapser.rs:234-259 */
{
    const uint64_t source = g->edge_node[0][edge], target = g->edge_node[1][edge];
    const uint64_t in_source = pg_degree_exact(g, source, 1);
    if (in_source == 0 || in_source > 2) return END;
    if (pg_degree_exact(g, target, 1) != 1 || pg_degree_exact(g, target, 0) != 2) return END;
    for (uint64_t e = g->node_next[0][target]; e != END; e = g->edge_next[0][e])
        if (g->edge_node[1][e] == source && g->edge_w[e] < g->edge_w[edge]) return e;
    return END;
}
static void decrease_weight(pgraph_t *g, uint64_t edge)                /* collapser.rs:261-273 */
{
    g->edge_w[edge] -= 1;
    if (g->edge_w[edge] > 0) return;
    pg_remove_edge(g, edge);
}
static void remove_single_with_ambiguity(pgraph_t *g, evec_t *to_remove, uint8_t *ambiguous)   /* collapser.rs:84-97 */
{
    if (to_remove->n) qsort(to_remove->v, to_remove->n, 8, cmp_desc);
    uint64_t last_node = g->n_nodes;
    for (size_t i = 0; i < to_remove->n; ++i) {
        last_node -= 1;
        ambiguous[to_remove->v[i]] = ambiguous[last_node];            /* copy_bit(last_node, node.index()) */
        pg_remove_node(g, to_remove->v[i]);
    }
    to_remove->n = 0;
}
static void contigs_from_vertex(pgraph_t *g, const labels_t *l, uint64_t v, uint8_t *ambiguous, evec_t *single_vertices,
                                strvec_t *contigs)                     /* collapser.rs:99-203 */
{
    str_t contig = {0, 0, 0};
    uint64_t current_vertex = v, current_edge_index, simple_loop_;
    uint64_t num_in = pg_degree_exact(g, current_vertex, 1), num_out = pg_degree_exact(g, current_vertex, 0);
    for (;;) {
        simple_loop_ = END;
        if (num_out == 0) {
            if (num_in == 0) evec_push(single_vertices, current_vertex);
            if (contig.n) strvec_push(contigs, contig.p, contig.n);
            free(contig.p);
            return;
        }
        current_edge_index = g->node_next[0][current_vertex];          /* first_edge(current_vertex, Outgoing) */
        if (ambiguous[current_vertex]) {
            if (contig.n) { strvec_push(contigs, contig.p, contig.n); contig.n = 0; }
        } else {
            int make_ambiguous = 0;
            if (num_in == 2 && num_out == 1) {
                if (self_loop(g, current_vertex) == END) {
                    simple_loop_ = simple_loop(g, current_edge_index);
                    if (simple_loop_ == END) make_ambiguous = 1;
                }
            } else if ((num_in == 1 && num_out == 2) || (num_in == 2 && num_out == 2)) {
                const uint64_t e = self_loop(g, current_vertex);
                if (e != END) current_edge_index = e; else make_ambiguous = 1;
            } else if ((num_in == 0 && num_out == 1) || (num_in == 1 && num_out == 1)) {
            } else {
                make_ambiguous = 1;
            }
            if (make_ambiguous) {
                ambiguous[current_vertex] = 1;
                if (contig.n) { strvec_push(contigs, contig.p, contig.n); contig.n = 0; }
            }
        }
        append_edge_name(&contig, l, g->edge_slot[current_edge_index], contig.n != 0);
        const uint64_t target = g->edge_node[1][current_edge_index];
        num_in = pg_degree_exact(g, target, 1);
        if (simple_loop_ != END) {
            append_edge_name(&contig, l, g->edge_slot[simple_loop_], 1);
            if (current_edge_index < simple_loop_) { decrease_weight(g, simple_loop_); decrease_weight(g, current_edge_index); }
            else { decrease_weight(g, current_edge_index); decrease_weight(g, simple_loop_); }
        } else {
            decrease_weight(g, current_edge_index);
        }
        num_out = pg_degree_exact(g, target, 0);
        if (pg_degree_exact(g, current_vertex, 1) == 0 && pg_degree_exact(g, current_vertex, 0) == 0) evec_push(single_vertices, current_vertex);
        current_vertex = target;
    }
}
/* Collapsable::collapse (collapser.rs:29-82) */
static void collapse(pgraph_t *g, labels_t *l, strvec_t *contigs)
{
    shrink(g, l);
    uint8_t *ambiguous = (uint8_t *)calloc(g->n_nodes ? g->n_nodes : 1, 1);
    evec_t single_vertices = {0, 0, 0}, externals = {0, 0, 0};
    for (;;) {
        for (;;) {
            externals.n = 0;
            for (uint64_t n = 0; n < g->n_nodes; ++n) if (g->node_next[1][n] == END) evec_push(&externals, n);
            if (externals.n == 0) break;
            for (size_t i = 0; i < externals.n; ++i) contigs_from_vertex(g, l, externals.v[i], ambiguous, &single_vertices, contigs);
            remove_single_with_ambiguity(g, &single_vertices, ambiguous);
        }
        if (g->n_nodes != 0) {
            const uint64_t node_in_cycle = last_scc_first_node(g);
            contigs_from_vertex(g, l, node_in_cycle, ambiguous, &single_vertices, contigs);
            remove_single_with_ambiguity(g, &single_vertices, ambiguous);
        } else {
            break;
        }
    }
    free(ambiguous); free(single_vertices.v); free(externals.v);
}

static void store_contigs(ko_graph *g, strvec_t *contigs)
{
    g->n_contigs = contigs->n;
    g->contig_off = (uint64_t *)xrealloc(NULL, (contigs->n + 1) * 8);
    size_t total = 0;
    for (size_t i = 0; i < contigs->n; ++i) total += strlen(contigs->v[i]);
    g->contig_seq = (uint8_t *)xrealloc(NULL, total + 8);
    size_t at = 0;
    for (size_t i = 0; i < contigs->n; ++i) {
        const size_t n = strlen(contigs->v[i]);
        g->contig_off[i] = at; memcpy(g->contig_seq + at, contigs->v[i], n); at += n;
        free(contigs->v[i]);
    }
    g->contig_off[contigs->n] = at;
    free(contigs->v); contigs->v = NULL; contigs->n = contigs->cap = 0;
}

/* stages run on the finished PtGraph before the result is read out, in the order given: 'd' = remove_dead_paths,
 * 'w' = remove_weak_edges(threshold), 's' = shrink */
static char g_stages[8] = "";
static uint32_t g_weak_threshold = 0;
static uint64_t g_genome_length = 0;          /* original_genome_length of standardize_edges ('e') */
void ko_set_genome_length(uint64_t n) { g_genome_length = n; }
void ko_set_post_build(const char *stages, uint32_t weak_threshold)
{
    size_t n = stages ? strlen(stages) : 0;
    if (n >= sizeof g_stages) n = sizeof g_stages - 1;
    memcpy(g_stages, stages ? stages : "", n); g_stages[n] = 0;
    g_weak_threshold = weak_threshold;
}
void ko_set_prune_dead_paths(int on) { ko_set_post_build(on ? "d" : "", 0); }

static ko_graph *finish(build_ctx *c)
{
    labels_t labels = {0, 0, 0};
    strvec_t contigs = {0, 0, 0};
    int collapsed = 0;
    for (const char *st = g_stages; *st; ++st) {
        if (*st == 'd') remove_dead_paths(&c->b.graph);
        else if (*st == 'w') remove_weak_edges(&c->b.graph, g_weak_threshold);
        else if (*st == 'c') standardize_contigs(&c->b.graph);
        else if (*st == 'e') standardize_edges(&c->b.graph, g_genome_length, K_SIZE, g_weak_threshold);
        else if (*st == 's' || *st == 'C') {
            if (!labels.bytes) {                       /* the post-pass of PtGraph::create (pt_graph.rs:339-343): slots -> edge format */
                labels.n = c->b.seqs.len;
                labels.bytes = (uint8_t **)calloc(labels.n, sizeof(uint8_t *));
                labels.len = (size_t *)calloc(labels.n, sizeof(size_t));
                const size_t stride = 1 + ceil_div(K_SIZE, CHARS_PER_CARRIER);
                for (size_t i = 1; i < labels.n; ++i) {
                    labels.bytes[i] = (uint8_t *)xrealloc(NULL, stride + 8);
                    labels.len[i] = ko_kmer_to_edge(c->b.seqs.data + i * c->b.seqs.slot, c->b.seqs.slot, labels.bytes[i]);
                }
            }
            if (*st == 's') shrink(&c->b.graph, &labels);
            else { collapse(&c->b.graph, &labels, &contigs); collapsed = 1; }
        }
    }
    ko_graph *g = (ko_graph *)calloc(1, sizeof *g);
    if (collapsed) store_contigs(g, &contigs);
    pgraph_t *p = &c->b.graph;
    g->n_nodes = p->n_nodes; g->n_edges = p->n_edges; g->read_bytes = c->total;
    g->label_stride = (uint32_t)(1 + ceil_div(K_SIZE, CHARS_PER_CARRIER));
    g->n_sequences = c->b.seqs.len;
    g->edge_src = (uint64_t *)xrealloc(NULL, p->n_edges * 8);
    g->edge_dst = (uint64_t *)xrealloc(NULL, p->n_edges * 8);
    g->edge_slot = (uint64_t *)xrealloc(NULL, p->n_edges * 8);
    g->edge_weight = (uint32_t *)xrealloc(NULL, p->n_edges * 4);
    g->edge_label = (uint8_t *)xrealloc(NULL, p->n_edges * (size_t)g->label_stride);
    for (uint64_t e = 0; e < p->n_edges; ++e) {
        g->edge_src[e] = p->edge_node[0][e];
        g->edge_dst[e] = p->edge_node[1][e];
        g->edge_slot[e] = p->edge_slot[e];
        g->edge_weight[e] = p->edge_w[e];
        ko_kmer_to_edge(c->b.seqs.data + p->edge_slot[e] * c->b.seqs.slot, c->b.seqs.slot,
                        g->edge_label + e * (size_t)g->label_stride);
    }
    if (labels.bytes) {      /* shrunk: labels of any length, handed out as ASCII (decompress_edge) with offsets */
        g->edge_seq_off = (uint64_t *)xrealloc(NULL, (p->n_edges + 1) * 8);
        size_t total = 0;
        for (uint64_t e = 0; e < p->n_edges; ++e) total += labels.len[p->edge_slot[e]] * 4;
        g->edge_seq = (uint8_t *)xrealloc(NULL, total + 8);
        size_t at = 0;
        for (uint64_t e = 0; e < p->n_edges; ++e) {
            g->edge_seq_off[e] = at;
            at += ko_decompress_edge(labels.bytes[p->edge_slot[e]], labels.len[p->edge_slot[e]], g->edge_seq + at);
        }
        g->edge_seq_off[p->n_edges] = at;
        labels_free(&labels);
    }
    pt_stats(p, &g->stats);
    if (c->with_gir) {       /* stats/collections.rs:190-208 */
        g->gir_node_count = c->gir.map.len;
        for (size_t i = 0; i < c->gir.n_out; ++i) g->gir_edge_count += c->gir.out[i].n;
    }
    return g;
}

/* Hand-made graphs, the shape of the reference's in-file tests (shrinker.rs:237-488, pruner.rs:259-393): `n_nodes` add_node
 * calls (more if an edge names a higher index, as PtGraph::from_edges does), then edge j = (src, dst, (EdgeSlice(slot), weight)),
 * then `stages` in order: 's' shrink, 'd' remove_dead_paths, 'w' remove_weak_edges(threshold), 'v' remove_single_vertices.
 * SEQUENCES slot i holds compress_edge(slot_ascii[i]) (slot 0 = scratch); slot_ascii may be NULL when no stage reads labels. */
int ko_run_from_edges(size_t n_nodes, const uint64_t *src, const uint64_t *dst, const uint64_t *slot, const uint32_t *w, size_t n_edges,
                      const char *const *slot_ascii, size_t n_slots, const char *stages, uint32_t threshold, size_t k, ko_graph **out)
{
    *out = NULL; g_err[0] = 0;
    ko_set_global_k_sizes(k);
    pgraph_t p; pg_init(&p);
    uint64_t need = n_nodes;
    for (size_t e = 0; e < n_edges; ++e) { if (src[e] + 1 > need) need = src[e] + 1; if (dst[e] + 1 > need) need = dst[e] + 1; }
    for (uint64_t n = 0; n < need; ++n) pg_add_node(&p);
    for (size_t e = 0; e < n_edges; ++e) pg_add_edge(&p, src[e], dst[e], slot ? slot[e] : 0, w[e]);
    labels_t l; l.n = slot_ascii ? n_slots : 0;
    l.bytes = (uint8_t **)calloc(l.n ? l.n : 1, sizeof(uint8_t *)); l.len = (size_t *)calloc(l.n ? l.n : 1, sizeof(size_t));
    for (size_t i = 1; i < l.n; ++i) {
        size_t n = strlen(slot_ascii[i]);
        l.bytes[i] = (uint8_t *)xrealloc(NULL, n / 4 + 8);
        l.len[i] = ko_compress_edge((const uint8_t *)slot_ascii[i], n, l.bytes[i]);
    }
    strvec_t contigs = {0, 0, 0};
    int collapsed = 0;
    for (const char *st = stages ? stages : ""; *st; ++st) {
        if (*st == 's') {
            if (!l.n) { snprintf(g_err, sizeof g_err, "oracle: shrink needs labels"); labels_free(&l); pg_free(&p); return KO_E_ARG; }
            shrink(&p, &l);
        }
        else if (*st == 'd') remove_dead_paths(&p);
        else if (*st == 'w') remove_weak_edges(&p, threshold);
        else if (*st == 'v') remove_single_vertices(&p);
        else if (*st == 'c') standardize_contigs(&p);
        else if (*st == 'e') standardize_edges(&p, g_genome_length, k, threshold);
        else if (*st == 'C') {
            if (!l.n) { snprintf(g_err, sizeof g_err, "oracle: collapse needs labels"); labels_free(&l); pg_free(&p); return KO_E_ARG; }
            collapse(&p, &l, &contigs); collapsed = 1;
        }
    }
    ko_graph *g = (ko_graph *)calloc(1, sizeof *g);
    if (collapsed) store_contigs(g, &contigs);
    g->n_nodes = p.n_nodes; g->n_edges = p.n_edges;
    g->edge_src = (uint64_t *)xrealloc(NULL, (p.n_edges + 1) * 8); g->edge_dst = (uint64_t *)xrealloc(NULL, (p.n_edges + 1) * 8);
    g->edge_slot = (uint64_t *)xrealloc(NULL, (p.n_edges + 1) * 8); g->edge_weight = (uint32_t *)xrealloc(NULL, (p.n_edges + 1) * 4);
    g->edge_seq_off = (uint64_t *)xrealloc(NULL, (p.n_edges + 1) * 8);
    size_t total = 0;
    for (uint64_t e = 0; e < p.n_edges && l.n; ++e) total += l.len[p.edge_slot[e]] * 4;
    g->edge_seq = (uint8_t *)xrealloc(NULL, total + 8);
    size_t at = 0;
    for (uint64_t e = 0; e < p.n_edges; ++e) {
        g->edge_src[e] = p.edge_node[0][e]; g->edge_dst[e] = p.edge_node[1][e]; g->edge_slot[e] = p.edge_slot[e]; g->edge_weight[e] = p.edge_w[e];
        g->edge_seq_off[e] = at;
        if (l.n && l.len[p.edge_slot[e]]) at += ko_decompress_edge(l.bytes[p.edge_slot[e]], l.len[p.edge_slot[e]], g->edge_seq + at);
    }
    g->edge_seq_off[p.n_edges] = at;
    pt_stats(&p, &g->stats);
    labels_free(&l); pg_free(&p);
    *out = g;
    return KO_OK;
}
int ko_shrink_from_edges(const uint64_t *src, const uint64_t *dst, const uint64_t *slot, const uint32_t *w, size_t n_edges,
                         const char *const *slot_ascii, size_t n_slots, size_t k, ko_graph **out)
{
    return ko_run_from_edges(0, src, dst, slot, w, n_edges, slot_ascii, n_slots, "s", 0, k, out);
}

int ko_build_files(const char *const *paths, size_t n_paths, int file_type, int reverse_complement,
                   size_t k, int with_gir, ko_graph **out)
{
    *out = NULL; g_err[0] = 0;
    if (k < 3 || k > 128) { snprintf(g_err, sizeof g_err, "oracle: k out of range"); return KO_E_ARG; }
    ko_set_global_k_sizes(k);
    build_ctx c; ctx_init(&c, reverse_complement, with_gir);
    uint64_t n_records = 0;
    int rc = scan_paths(paths, n_paths, file_type, build_read_cb, &c, &n_records);
    if (!rc) *out = finish(&c);
    ctx_free(&c);
    return rc;
}

static int build_bfc_cb(void *vctx, const uint8_t *kmer, size_t len, uint32_t weight)
{
    build_ctx *c = (build_ctx *)vctx;
    return add_read_bfc(&c->b, kmer, len, weight, c->rc);
}

/* Build::create with InputFileType::BFCounter (builder.rs:50-52,79-115) */
int ko_build_bfc(const char *const *paths, size_t n_paths, int reverse_complement, uint32_t threshold, size_t k, ko_graph **out)
{
    *out = NULL; g_err[0] = 0;
    if (k < 3 || k > 128) { snprintf(g_err, sizeof g_err, "oracle: k out of range"); return KO_E_ARG; }
    ko_set_global_k_sizes(k);
    char (*resolved)[PATH_MAX] = (char (*)[PATH_MAX])xrealloc(NULL, (n_paths ? n_paths : 1) * PATH_MAX);
    int rc = KO_OK;
    for (size_t i = 0; i < n_paths && !rc; ++i) rc = check_file(paths[i], resolved[i]);
    build_ctx c; ctx_init(&c, reverse_complement, 0);
    uint64_t total = 0;
    for (size_t i = 0; i < n_paths && !rc; ++i) {
        linebuf_t lb;
        rc = slurp(resolved[i], &lb);
        if (rc) break;
        rc = scan_bfc(&lb, threshold, build_bfc_cb, &c, &total);
        free(lb.buf);
    }
    c.total = total;
    if (!rc) *out = finish(&c);
    ctx_free(&c);
    free(resolved);
    return rc;
}

int ko_build_ascii(const uint8_t *reads, size_t n_reads, size_t read_len, int reverse_complement,
                   size_t k, int with_gir, ko_graph **out)
{
    *out = NULL; g_err[0] = 0;
    if (k < 3 || k > 128) { snprintf(g_err, sizeof g_err, "oracle: k out of range"); return KO_E_ARG; }
    ko_set_global_k_sizes(k);
    build_ctx c; ctx_init(&c, reverse_complement, with_gir);
    int rc = KO_OK;
    for (size_t r = 0; r < n_reads && !rc; ++r) rc = build_read_cb(&c, reads + r * read_len, read_len);
    if (!rc) *out = finish(&c);
    ctx_free(&c);
    return rc;
}

void ko_graph_free(ko_graph *g)
{
    if (!g) return;
    free(g->edge_src); free(g->edge_dst); free(g->edge_slot); free(g->edge_weight); free(g->edge_label);
    free(g->edge_seq_off); free(g->edge_seq); free(g->contig_off); free(g->contig_seq);
    free(g);
}

/* ------------------------------ record scan only ---------------------------------- */
typedef struct { ko_reads *r; size_t cap_seq, cap_off; } scan_ctx;
static int scan_read_cb(void *vctx, const uint8_t *seq, size_t len)
{
    scan_ctx *c = (scan_ctx *)vctx;
    if (!all_acgt(seq, len)) return KO_OK;
    ko_reads *r = c->r;
    if (r->read_bytes + len > c->cap_seq) { c->cap_seq = (r->read_bytes + len) * 2 + 1024; r->seq = (uint8_t *)xrealloc(r->seq, c->cap_seq); }
    if (r->n_accepted + 2 > c->cap_off) { c->cap_off = (r->n_accepted + 2) * 2; r->off = (uint64_t *)xrealloc(r->off, c->cap_off * 8); }
    memcpy(r->seq + r->read_bytes, seq, len);
    r->off[r->n_accepted] = r->read_bytes;
    r->read_bytes += len;
    r->off[++r->n_accepted] = r->read_bytes;
    return KO_OK;
}
int ko_scan_files(const char *const *paths, size_t n_paths, int file_type, ko_reads **out)
{
    g_err[0] = 0;
    scan_ctx c; memset(&c, 0, sizeof c);
    c.r = (ko_reads *)calloc(1, sizeof(ko_reads));
    c.r->off = (uint64_t *)xrealloc(NULL, 16); c.r->off[0] = 0; c.cap_off = 2;
    int rc = scan_paths(paths, n_paths, file_type, scan_read_cb, &c, &c.r->n_records);
    if (rc) { ko_reads_free(c.r); *out = NULL; return rc; }
    *out = c.r;
    return KO_OK;
}
void ko_reads_free(ko_reads *r) { if (!r) return; free(r->seq); free(r->off); free(r); }

/* ============================ synthetic workload =================================== */
/* Deterministic generator shared (by definition, not by code) with the device generator
 * in katome_amd/csrc/synth.hip -- DESIGN.md "Synthetic workload".                       */
uint64_t ko_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void ko_synth_reads(uint64_t first_read, uint64_t n_reads, uint32_t read_len, uint64_t genome_len,
                    double err_rate, uint32_t n_inject_percent, uint8_t *out_ascii)
{
    static const uint8_t sym[4] = {'A', 'C', 'G', 'T'};
    const uint64_t BG = ko_splitmix64(0x6B61746F6D650001ull), BR = ko_splitmix64(0x6B61746F6D650002ull),
                   BE = ko_splitmix64(0x6B61746F6D650003ull), BN = ko_splitmix64(0x6B61746F6D650004ull);
    const uint64_t thr = (uint64_t)(err_rate * 16777216.0);
    const uint64_t L = read_len;
    for (uint64_t i = 0; i < n_reads; ++i) {
        uint64_t r = first_read + i;
        uint64_t start = ko_splitmix64(BR + 2 * r) % (genome_len - L + 1);
        uint64_t strand = ko_splitmix64(BR + 2 * r + 1) & 1;
        uint8_t *row = out_ascii + i * L;
        for (uint64_t j = 0; j < L; ++j) {
            uint64_t b = strand ? 3 - (ko_splitmix64(BG + start + (L - 1 - j)) & 3) : (ko_splitmix64(BG + start + j) & 3);
            uint64_t x = ko_splitmix64(BE + r * L + j);
            if ((x >> 40) < thr) b = (b + 1 + ((x & 0xFFFF) % 3)) & 3;
            row[j] = sym[b];
        }
        uint64_t u = ko_splitmix64(BN + r);
        if (n_inject_percent && (u % 100) < n_inject_percent) row[(u >> 32) % L] = 'N';
    }
}

/* ============================ pruner.rs:84-93 (PtGraph) ============================= */
