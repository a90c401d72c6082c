"""ctypes wrapper around the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under katome_amd/ does.  The oracle restates the reference's `build`
stage in plain C (oracle/katome_oracle.c, each function citing the reference file:line).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libkatome_oracle.so")
    src = os.path.join(_HERE, "katome_oracle.c")
    hdr = os.path.join(_HERE, "katome_oracle.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-g", "-fPIC", "-std=c99", "-shared", "-o", so, src])
    return so


class KoStats(C.Structure):
    _fields_ = [("node_count", C.c_uint64), ("edge_count", C.c_uint64),
                ("max_edge_weight", C.c_uint32), ("avg_edge_weight", C.c_double),
                ("max_in_degree", C.c_uint64), ("max_out_degree", C.c_uint64),
                ("avg_out_degree", C.c_double),
                ("incoming_vert_count", C.c_uint64), ("outgoing_vert_count", C.c_uint64)]


class KoGraph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_edges", C.c_uint64), ("read_bytes", C.c_uint64),
                ("edge_src", C.POINTER(C.c_uint64)), ("edge_dst", C.POINTER(C.c_uint64)),
                ("edge_weight", C.POINTER(C.c_uint32)), ("edge_slot", C.POINTER(C.c_uint64)),
                ("edge_label", C.POINTER(C.c_uint8)), ("label_stride", C.c_uint32),
                ("n_sequences", C.c_uint64), ("stats", KoStats),
                ("gir_node_count", C.c_uint64), ("gir_edge_count", C.c_uint64),
                ("edge_seq_off", C.POINTER(C.c_uint64)), ("edge_seq", C.POINTER(C.c_uint8)),
                ("n_contigs", C.c_uint64), ("contig_off", C.POINTER(C.c_uint64)), ("contig_seq", C.POINTER(C.c_uint8))]


class KoReads(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_accepted", C.c_uint64), ("read_bytes", C.c_uint64),
                ("seq", C.POINTER(C.c_uint8)), ("off", C.POINTER(C.c_uint64))]


ERRORS = {0: "OK", -1: "E_PATH", -2: "E_IS_DIR", -3: "E_NOT_EXIST", -4: "E_OPEN", -5: "E_PARSE",
          -6: "E_SHORT_READ", -7: "E_ARG"}


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p = C.POINTER(C.c_uint8)
        L.ko_set_global_k_sizes.argtypes = [C.c_size_t]
        for f in ("ko_k_size", "ko_k1_size", "ko_compressed_k1_size"):
            getattr(L, f).restype = C.c_size_t
        L.ko_encode_fasta_symbol.argtypes = [C.c_uint8, C.c_uint8]
        L.ko_encode_fasta_symbol.restype = C.c_uint8
        for f in ("ko_compress_node", "ko_compress_kmer", "ko_compress_edge", "ko_decompress_edge",
                  "ko_decompress_node", "ko_decompress_kmer", "ko_kmer_to_edge"):
            getattr(L, f).argtypes = [C.c_char_p, C.c_size_t, u8p]
            getattr(L, f).restype = C.c_size_t
        L.ko_compress_kmer_with_rev_compl.argtypes = [C.c_char_p, C.c_size_t, u8p, u8p]
        L.ko_compress_kmer_with_rev_compl.restype = C.c_size_t
        L.ko_reverse_compressed_node.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, u8p]
        L.ko_shift_left_bit_array.argtypes = [u8p, C.c_size_t, C.c_size_t]
        L.ko_shift_right_bit_array.argtypes = [u8p, C.c_size_t, C.c_size_t]
        L.ko_add_char_to_edge.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, u8p]
        L.ko_add_char_to_edge.restype = C.c_size_t
        L.ko_change_last_char_in_edge.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, u8p]
        L.ko_change_last_char_in_edge.restype = C.c_size_t
        L.ko_extend_edge.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, u8p]
        L.ko_extend_edge.restype = C.c_size_t
        L.ko_decompress_char.argtypes = [C.c_uint8, C.c_size_t]
        L.ko_decompress_char.restype = C.c_char
        L.ko_decode_compressed_chunk.argtypes = [C.c_uint8, u8p]
        L.ko_build_files.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.c_int, C.c_size_t,
                                     C.c_int, C.POINTER(C.POINTER(KoGraph))]
        L.ko_build_bfc.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.c_uint32, C.c_size_t,
                                   C.POINTER(C.POINTER(KoGraph))]
        L.ko_build_ascii.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_int,
                                     C.POINTER(C.POINTER(KoGraph))]
        L.ko_graph_free.argtypes = [C.POINTER(KoGraph)]
        L.ko_set_post_build.argtypes = [C.c_char_p, C.c_uint32]
        L.ko_set_post_build.restype = None
        L.ko_last_prune_passes.restype = C.c_uint64
        L.ko_last_error.restype = C.c_char_p
        L.ko_scan_files.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_int, C.POINTER(C.POINTER(KoReads))]
        L.ko_reads_free.argtypes = [C.POINTER(KoReads)]
        L.ko_splitmix64.argtypes = [C.c_uint64]
        L.ko_splitmix64.restype = C.c_uint64
        L.ko_synth_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_double, C.c_uint32,
                                     C.c_void_p]
        _LIB = L
    return _LIB


def _out(n):
    return (C.c_uint8 * n)()


def _call_bytes(fn, data, cap=None, *extra):
    buf = _out(cap or (len(data) * 4 + 16))
    n = fn(bytes(data), len(data), *extra, buf)
    return bytes(buf[:n])


# ---- codec -------------------------------------------------------------------------------
def set_k(k):
    lib().ko_set_global_k_sizes(k)


def encode_fasta_symbol(sym, carrier=0):
    return lib().ko_encode_fasta_symbol(sym, carrier)


def compress_node(s):
    return _call_bytes(lib().ko_compress_node, s)


def compress_kmer(s):
    return _call_bytes(lib().ko_compress_kmer, s)


def compress_kmer_with_rev_compl(s):
    a, b = _out(len(s) + 16), _out(len(s) + 16)
    n = lib().ko_compress_kmer_with_rev_compl(bytes(s), len(s), a, b)
    return bytes(a[:n]), bytes(b[:n])


def reverse_compressed_node(v, remainder):
    v = bytes(v)
    buf = _out(len(v))
    lib().ko_reverse_compressed_node(v, len(v), remainder, buf)
    return bytes(buf)


def shift_right_bit_array(v, s):
    buf = (C.c_uint8 * len(v))(*v)
    lib().ko_shift_right_bit_array(buf, len(v), s)
    return bytes(buf)


def shift_left_bit_array(v, s):
    buf = (C.c_uint8 * len(v))(*v)
    lib().ko_shift_left_bit_array(buf, len(v), s)
    return bytes(buf)


def compress_edge(s):
    return _call_bytes(lib().ko_compress_edge, s)


def decompress_edge(v):
    return _call_bytes(lib().ko_decompress_edge, v)


def decompress_kmer(v):
    return _call_bytes(lib().ko_decompress_kmer, v)


def kmer_to_edge(v):
    return _call_bytes(lib().ko_kmer_to_edge, v)


def add_char_to_edge(v, ch):
    buf = _out(len(v) + 2)
    n = lib().ko_add_char_to_edge(bytes(v), len(v), ch, buf)
    return bytes(buf[:n])


def change_last_char_in_edge(v, ch):
    buf = _out(len(v) + 2)
    n = lib().ko_change_last_char_in_edge(bytes(v), len(v), ch, buf)
    return bytes(buf[:n])


def extend_edge(v, w):
    buf = _out(len(v) + len(w) + 4)
    n = lib().ko_extend_edge(bytes(v), len(v), bytes(w), len(w), buf)
    return bytes(buf[:n])


def decompress_char(chunk, padding):
    return lib().ko_decompress_char(chunk, padding).decode()


def decode_compressed_chunk(chunk):
    buf = _out(4)
    lib().ko_decode_compressed_chunk(chunk, buf)
    return bytes(buf)


# ---- build -------------------------------------------------------------------------------
class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERRORS.get(code, code), msg))
        self.code = code
        self.name = ERRORS.get(code, str(code))


class OracleGraph:
    """Result of the restated PtGraph::create, copied into numpy arrays."""

    def __init__(self, gp, k):
        g = gp.contents
        ne = g.n_edges
        self.k = k
        self.n_nodes, self.n_edges, self.read_bytes = g.n_nodes, ne, g.read_bytes
        self.label_stride = g.label_stride
        self.n_sequences = g.n_sequences
        self.edge_src = np.ctypeslib.as_array(g.edge_src, (ne,)).copy() if ne else np.zeros(0, np.uint64)
        self.edge_dst = np.ctypeslib.as_array(g.edge_dst, (ne,)).copy() if ne else np.zeros(0, np.uint64)
        self.edge_weight = np.ctypeslib.as_array(g.edge_weight, (ne,)).copy() if ne else np.zeros(0, np.uint32)
        self.edge_slot = np.ctypeslib.as_array(g.edge_slot, (ne,)).copy() if ne else np.zeros(0, np.uint64)
        self.edge_label = (np.ctypeslib.as_array(g.edge_label, (ne, g.label_stride)).copy()
                           if ne and bool(g.edge_label) else np.zeros((0, g.label_stride), np.uint8))
        s = g.stats
        self.stats = dict(node_count=s.node_count, edge_count=s.edge_count, max_edge_weight=s.max_edge_weight,
                          avg_edge_weight=s.avg_edge_weight, max_in_degree=s.max_in_degree,
                          max_out_degree=s.max_out_degree, avg_out_degree=s.avg_out_degree,
                          incoming_vert_count=s.incoming_vert_count, outgoing_vert_count=s.outgoing_vert_count)
        self.gir_counts = (g.gir_node_count, g.gir_edge_count)
        self.edge_seq = None                 # after a shrink stage: every edge's whole sequence
        if bool(g.edge_seq_off):
            off = np.ctypeslib.as_array(g.edge_seq_off, (ne + 1,)).copy()
            raw = bytes(np.ctypeslib.as_array(g.edge_seq, (max(int(off[-1]), 1),))[:int(off[-1])])
            self.edge_seq = [raw[int(off[i]):int(off[i + 1])].decode() for i in range(ne)]

        self.collapsed = None                # after a collapse stage: the serialized contigs, in order
        if bool(g.contig_off):
            nc = g.n_contigs
            off = np.ctypeslib.as_array(g.contig_off, (nc + 1,)).copy()
            raw = bytes(np.ctypeslib.as_array(g.contig_seq, (max(int(off[-1]), 1),))[:int(off[-1])])
            self.collapsed = [raw[int(off[i]):int(off[i + 1])].decode() for i in range(nc)]

    def contigs(self):
        """after shrink: sorted (sequence, weight) of every edge; the endpoints are its first and last k-1 bases"""
        return sorted(zip(self.edge_seq, (int(w) for w in self.edge_weight)))

    def kmer_strings(self):
        """decompress_edge of every label -> list of ASCII k-mers (small graphs only)."""
        return [decompress_edge(bytes(row)).decode() for row in self.edge_label]

    def multiset(self):
        """sorted list of (k-mer string, weight)"""
        return sorted(zip(self.kmer_strings(), (int(w) for w in self.edge_weight)))


def _paths(paths):
    arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
    return arr


def set_genome_length(n):
    """original_genome_length for the 'e' (standardize_edges) stage"""
    lib().ko_set_genome_length.argtypes = [C.c_uint64]
    lib().ko_set_genome_length(int(n))


def _stages(remove_weak_edges, remove_dead_paths, stages):
    """post-build stages on the PtGraph, in order: 'w' = Clean::remove_weak_edges(threshold) (pruner.rs:84-93),
    'd' = Prunable::remove_dead_paths (pruner.rs:36-82), 's' = Shrinkable::shrink (shrinker.rs:165-176),
    'c' = standardize_contigs, 'e' = standardize_edges(genome length via set_genome_length, threshold = remove_weak_edges);
    default: weak edges first if both are asked for"""
    if stages is None:
        stages = ("w" if remove_weak_edges is not None else "") + ("d" if remove_dead_paths else "")
    lib().ko_set_post_build(stages.encode(), int(remove_weak_edges or 0))


def build_files(paths, k, reverse_complement=False, file_type=1, with_gir=False, remove_weak_edges=None,
                remove_dead_paths=False, stages=None):
    gp = C.POINTER(KoGraph)()
    _stages(remove_weak_edges, remove_dead_paths, stages)
    rc = lib().ko_build_files(_paths(paths), len(paths), file_type, int(reverse_complement), k, int(with_gir),
                              C.byref(gp))
    lib().ko_set_post_build(b"", 0)
    if rc:
        raise OracleError(rc, lib().ko_last_error().decode())
    try:
        return OracleGraph(gp, k)
    finally:
        lib().ko_graph_free(gp)


def build_bfc(paths, k, reverse_complement=False, threshold=0, remove_dead_paths=False, stages=None):
    gp = C.POINTER(KoGraph)()
    _stages(None, remove_dead_paths, stages)
    rc = lib().ko_build_bfc(_paths(paths), len(paths), int(reverse_complement), threshold, k, C.byref(gp))
    lib().ko_set_post_build(b"", 0)
    if rc:
        raise OracleError(rc, lib().ko_last_error().decode())
    try:
        return OracleGraph(gp, k)
    finally:
        lib().ko_graph_free(gp)


def build_ascii(reads, k, reverse_complement=False, with_gir=False, remove_dead_paths=False, remove_weak_edges=None,
                stages=None):
    """reads: uint8 array [n_reads, read_len] of ASCII codes."""
    reads = np.ascontiguousarray(reads, dtype=np.uint8)
    gp = C.POINTER(KoGraph)()
    _stages(remove_weak_edges, remove_dead_paths, stages)
    rc = lib().ko_build_ascii(reads.ctypes.data, reads.shape[0], reads.shape[1], int(reverse_complement), k,
                              int(with_gir), C.byref(gp))
    lib().ko_set_post_build(b"", 0)
    if rc:
        raise OracleError(rc, lib().ko_last_error().decode())
    try:
        return OracleGraph(gp, k)
    finally:
        lib().ko_graph_free(gp)


def shrink_from_edges(edges, slot_ascii, k):
    """PtGraph::from_edges(edges) then shrink(): edges = [(src, dst, slot, weight)], slot_ascii[i] = the ASCII edge in
    SEQUENCES slot i (slot 0 unused) -> OracleGraph with edge_seq"""
    n = len(edges)
    arr = lambda vals, t: (t * max(n, 1))(*vals)
    slots = (C.c_char_p * len(slot_ascii))(*[x.encode() if x else b"" for x in slot_ascii])
    gp = C.POINTER(KoGraph)()
    L = lib()
    L.ko_shrink_from_edges.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_size_t, C.POINTER(C.POINTER(KoGraph))]
    rc = L.ko_shrink_from_edges(arr([e[0] for e in edges], C.c_uint64), arr([e[1] for e in edges], C.c_uint64),
                                arr([e[2] for e in edges], C.c_uint64), arr([e[3] for e in edges], C.c_uint32), n,
                                slots, len(slot_ascii), k, C.byref(gp))
    if rc:
        raise OracleError(rc, lib().ko_last_error().decode())
    try:
        return OracleGraph(gp, k)
    finally:
        lib().ko_graph_free(gp)


def run_from_edges(n_nodes, edges, stages, threshold=0, k=40, slot_ascii=None):
    """hand-made PtGraph (n_nodes add_node calls; edges = [(src, dst, weight)] or, with slot_ascii (slot 0 unused),
    [(src, dst, weight, slot)]) through `stages` -> OracleGraph"""
    n = len(edges)
    arr = lambda vals, t: (t * max(n, 1))(*vals)
    gp = C.POINTER(KoGraph)()
    L = lib()
    L.ko_run_from_edges.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                    C.c_char_p, C.c_uint32, C.c_size_t, C.POINTER(C.POINTER(KoGraph))]
    slots = (C.c_char_p * len(slot_ascii))(*[x.encode() if x else b"" for x in slot_ascii]) if slot_ascii else None
    slot_arr = arr([e[3] for e in edges], C.c_uint64) if slot_ascii else None
    rc = L.ko_run_from_edges(n_nodes, arr([e[0] for e in edges], C.c_uint64), arr([e[1] for e in edges], C.c_uint64), slot_arr,
                             arr([e[2] for e in edges], C.c_uint32), n, slots, len(slot_ascii) if slot_ascii else 0,
                             stages.encode(), threshold, k, C.byref(gp))
    if rc:
        raise OracleError(rc, lib().ko_last_error().decode())
    try:
        return OracleGraph(gp, k)
    finally:
        lib().ko_graph_free(gp)


def scan_files(paths, file_type=1):
    rp = C.POINTER(KoReads)()
    rc = lib().ko_scan_files(_paths(paths), len(paths), file_type, C.byref(rp))
    if rc:
        raise OracleError(rc, lib().ko_last_error().decode())
    r = rp.contents
    off = np.ctypeslib.as_array(r.off, (r.n_accepted + 1,)).copy() if bool(r.off) else np.zeros(1, np.uint64)
    seq = np.ctypeslib.as_array(r.seq, (max(r.read_bytes, 1),)).copy()[:r.read_bytes] if bool(r.seq) else np.zeros(0, np.uint8)
    out = dict(n_records=r.n_records, n_accepted=r.n_accepted, read_bytes=r.read_bytes, seq=seq, off=off)
    lib().ko_reads_free(rp)
    return out


def splitmix64(x):
    return lib().ko_splitmix64(x)


def synth_reads(first_read, n_reads, read_len, genome_len, err_rate, n_inject_percent=0):
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    lib().ko_synth_reads(first_read, n_reads, read_len, genome_len, err_rate, n_inject_percent, out.ctypes.data)
    return out


def last_prune_passes():
    """iterations of remove_dead_paths' outer loop in the last pruned build (the final empty one included)"""
    return int(lib().ko_last_prune_passes())
