"""Host-side mirror of the reference's build interface for the GPU path.

Mirrors (names, argument meaning, error behaviour):
  * `Config<P>` and `InputFileType`          -- reference src/katome/config.rs:5-37
  * `set_global_k_sizes`                     -- prelude.rs:34-43
  * `Build::create(input_files, ft, reverse_complement, minimal_weight_threshold) -> (Self, usize)`
                                             -- algorithms/builder.rs:42-54
  * `Stats<CollectionStats>::stats`          -- stats/collections.rs:38-89,137-168
The reference panics on every failure (builder.rs:62,67,71,148,153; pt_graph.rs:278); here a
non-zero status of the C ABI is re-raised as `KatomePanic` carrying the same message, which is
what the Rust shim of INTEGRATION.md does with `panic!`.
"""
import ctypes as C
import enum
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib

# prelude.rs:21-25 -- the reference keeps k in `static mut` globals
K_SIZE = 40
K1_SIZE = 39
COMPRESSED_K1_SIZE = 10


class KatomePanic(RuntimeError):
    """A failure the reference reports by panicking."""

    def __init__(self, status, message):
        super().__init__("%s: %s" % (_lib.STATUS.get(status, status), message))
        self.status = status
        self.name = _lib.STATUS.get(status, str(status))
        self.message = message


def _check(status):
    if status != 0:
        raise KatomePanic(status, _lib.last_error())


def set_global_k_sizes(k_size: int):
    """prelude.rs:34-43"""
    global K_SIZE, K1_SIZE, COMPRESSED_K1_SIZE
    assert k_size > 1
    K_SIZE = k_size
    K1_SIZE = k_size - 1
    COMPRESSED_K1_SIZE = (K1_SIZE + 3) // 4


class InputFileType(enum.IntEnum):
    """config.rs:5-14; parsing is case-insensitive like utils.rs:15-19"""
    Fasta = 0
    Fastq = 1
    BFCounter = 2

    @classmethod
    def parse(cls, text: str) -> "InputFileType":
        for m in cls:
            if m.name.lower() == text.lower():
                return m
        raise ValueError("Bad value %r for InputFileType" % text)


@dataclass
class Config:
    """config.rs:18-37"""
    input_files: List[str]
    input_file_type: InputFileType
    output_file: str = ""
    original_genome_length: int = 0
    minimal_weight_threshold: int = 0
    k_mer_size: int = 40
    reverse_complement: bool = True


def _round2(x):
    # stats/collections.rs:84-89 (f64::round = half away from zero; values here are >= 0)
    return None if x is None else float(np.floor(x * 100.0 + 0.5) / 100.0)


@dataclass
class CollectionStats:
    """stats/collections.rs:38-57; equality ignores capacity and rounds the averages to 2 dp (71-82)"""
    node_count: int = 0
    edge_count: int = 0
    max_edge_weight: Optional[int] = None
    avg_edge_weight: Optional[float] = None
    max_in_degree: Optional[int] = None
    max_out_degree: Optional[int] = None
    avg_out_degree: Optional[float] = None
    incoming_vert_count: Optional[int] = None
    outgoing_vert_count: Optional[int] = None
    capacity: tuple = field(default=(0, None), compare=False)

    def __eq__(self, other):
        return (self.node_count == other.node_count and self.edge_count == other.edge_count and
                self.max_edge_weight == other.max_edge_weight and
                _round2(self.avg_edge_weight) == _round2(other.avg_edge_weight) and
                self.max_in_degree == other.max_in_degree and self.max_out_degree == other.max_out_degree and
                _round2(self.avg_out_degree) == _round2(other.avg_out_degree) and
                self.incoming_vert_count == other.incoming_vert_count and
                self.outgoing_vert_count == other.outgoing_vert_count)

    @classmethod
    def with_counts(cls, node_count, edge_count):
        return cls(node_count=node_count, edge_count=edge_count)


FLAG_FIRST_SEEN_ORDER = 1
FLAG_REMOVE_DEAD_PATHS = 2
FLAG_RANKS_SHARE_DEVICE = 4


def make_settings(k, file_type=InputFileType.Fastq, reverse_complement=False, min_weight=0, device=0,
                  table_slots_hint=0, first_seen_order=False, remove_dead_paths=False, n_devices=1,
                  ranks_share_device=False):
    """n_devices: GPUs of this node to build on (device .. device+n-1; one rank per GPU inside the call);
    ranks_share_device: all those ranks on `device` -- the rehearsal of the sharded route on a one-GPU box"""
    s = _lib.Settings()
    s.k = k
    s.file_type = int(file_type)
    s.reverse_complement = 1 if reverse_complement else 0
    s.flags = ((FLAG_FIRST_SEEN_ORDER if first_seen_order else 0) | (FLAG_REMOVE_DEAD_PATHS if remove_dead_paths else 0) |
               (FLAG_RANKS_SHARE_DEVICE if ranks_share_device else 0))
    s.min_weight = min_weight
    s.device = device
    s.table_slots_hint = table_slots_hint
    s.n_devices = n_devices
    return s


def _paths(paths):
    return (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])


class GpuGraph:
    """The collection the GPU build produces: what `Convert<GpuGIR> for PtGraph` consumes.

    Arrays (numpy views of the C result, which lives as long as any of them does): edge_src/edge_dst (dense node ids),
    edge_weight (u32), edge_label ([n_edges, 1+ceil(k/4)] compress_edge format), edge_key / node_key (packed k-mers /
    (k-1)-mers, [n, key_words] u64, most significant word first).
    """

    class _Owner:
        """frees the katome_graph when the last array that views it is gone"""

        def __init__(self, gptr):
            self.gptr = gptr

        def __del__(self):
            try:
                _lib.lib().katome_graph_free(self.gptr)
            except Exception:       # interpreter shutdown
                pass

    def __init__(self, gptr):
        g = gptr.contents
        owner = GpuGraph._Owner(gptr)
        ne, nn, nw = g.n_edges, g.n_nodes, g.key_words
        self.k = g.k
        self.n_nodes, self.n_edges, self.read_bytes = nn, ne, g.read_bytes
        self.key_words, self.label_stride = nw, g.label_stride

        def arr(ptr, shape, dtype):
            n = int(np.prod(shape))
            if n == 0:
                return np.zeros(shape, dtype)
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents))
            buf._owner = owner                      # numpy keeps `buf` as the arrays' base, `buf` keeps the C result
            return np.frombuffer(buf, dtype=dtype).reshape(shape)

        self.edge_src = arr(g.edge_src, (ne,), np.uint64)
        self.edge_dst = arr(g.edge_dst, (ne,), np.uint64)
        self.edge_weight = arr(g.edge_weight, (ne,), np.uint32)
        self.edge_label = arr(g.edge_label, (ne, g.label_stride), np.uint8)
        self.edge_key = arr(g.edge_key, (ne, nw), np.uint64)
        self.node_key = arr(g.node_key, (nn, nw), np.uint64)
        # first-seen index each edge had before remove_dead_paths / remove_weak_edges re-numbered it (None: age == index)
        self.edge_age = arr(g.edge_age, (ne,), np.uint32) if bool(g.edge_age) else None
        self._gptr, self._owner, self._stats = gptr, owner, None      # (stats: on the first stats() call, as in the reference)

    # ---- Build::create (builder.rs:42-54) ----------------------------------------------------
    @classmethod
    def create(cls, input_files, ft, reverse_complement, minimal_weight_threshold=0, device=0, first_seen_order=False,
               remove_dead_paths=False, stages=None, original_genome_length=0, n_devices=1, ranks_share_device=False):
        """-> (GpuGraph, number_of_read_bytes); uses the global k set by set_global_k_sizes.
        stages: stages of assemble_with_graph to run on the device after the build, e.g. "dcwced" = everything before
        collapse (d remove_dead_paths, c standardize_contigs, w remove_weak_edges(minimal_weight_threshold),
        e standardize_edges(original_genome_length, k, minimal_weight_threshold)); needs first_seen_order.
        first_seen_order: number edges and nodes as the reference's petgraph does (order of first insertion).
        remove_dead_paths: also run Prunable::remove_dead_paths (pruner.rs:36-82) as assemble() does next
        (asm/basic_assembler.rs:58-62); needs first_seen_order."""
        s = make_settings(K_SIZE, ft, reverse_complement, minimal_weight_threshold, device,
                          first_seen_order=first_seen_order, remove_dead_paths=remove_dead_paths, n_devices=n_devices,
                          ranks_share_device=ranks_share_device)
        gp = C.POINTER(_lib.Graph)()
        if stages:
            _check(_lib.lib().katome_build_files_staged(C.byref(s), _paths(input_files), len(input_files), stages.encode(),
                                                        original_genome_length, C.byref(gp)))
        else:
            _check(_lib.lib().katome_build_files(C.byref(s), _paths(input_files), len(input_files), C.byref(gp)))
        g = cls(gp)                                   # the arrays view the C result and free it with their last reference
        return g, g.read_bytes

    @classmethod
    def create_from_packed(cls, packed, n_reads, read_len, skip=None, reverse_complement=False, device=0, k=None,
                           first_seen_order=False, remove_dead_paths=False, n_devices=1, ranks_share_device=False,
                           table_slots_hint=0):
        """Same build from 2-bit packed reads (numpy uint8), the synthetic-workload entry."""
        s = make_settings(K_SIZE if k is None else k, InputFileType.Fastq, reverse_complement, 0, device,
                          table_slots_hint=table_slots_hint, first_seen_order=first_seen_order,
                          remove_dead_paths=remove_dead_paths, n_devices=n_devices, ranks_share_device=ranks_share_device)
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        skip_p = None
        if skip is not None:
            skip = np.ascontiguousarray(skip, dtype=np.uint8)
            skip_p = skip.ctypes.data
        gp = C.POINTER(_lib.Graph)()
        _check(_lib.lib().katome_build_packed(C.byref(s), packed.ctypes.data, n_reads, read_len, skip_p, C.byref(gp)))
        g = cls(gp)
        return g, g.read_bytes

    # ---- Stats<CollectionStats> (stats/collections.rs:137-168) ---------------------------------
    def stats(self) -> CollectionStats:
        # a pass over the host arrays (degrees, weights): Stats::stats is its own call in the reference too, not part of create
        if self._stats is None:
            st = _lib.Stats()
            _check(_lib.lib().katome_graph_stats(self._gptr, C.byref(st)))
            self._stats = CollectionStats(
                node_count=st.node_count, edge_count=st.edge_count, max_edge_weight=st.max_edge_weight,
                avg_edge_weight=st.avg_edge_weight, max_in_degree=st.max_in_degree, max_out_degree=st.max_out_degree,
                avg_out_degree=st.avg_out_degree, incoming_vert_count=st.incoming_vert_count,
                outgoing_vert_count=st.outgoing_vert_count, capacity=(self.n_nodes, self.n_edges))
        return self._stats

    # ---- helpers for parity checks --------------------------------------------------------------
    def key_ints(self, which="edge"):
        """packed keys as Python ints"""
        a = self.edge_key if which == "edge" else self.node_key
        if self.key_words == 1:
            return [int(x) for x in a[:, 0]]
        return [(int(h) << 64) | int(l) for h, l in a]

    def kmer_strings(self):
        k = self.k
        return [_int_to_kmer(v, k) for v in self.key_ints("edge")]

    def multiset(self):
        """sorted list of (k-mer string, weight) -- the parity observable"""
        return sorted(zip(self.kmer_strings(), (int(w) for w in self.edge_weight)))


class GpuContigs:
    """The build followed by Shrinkable::shrink (shrinker.rs:165-209): one edge per maximal straight path.

    edge_seq: the paths as ACGT strings (decoded from the compress_edge labels); edge_weight: weight of each path's
    first k-mer; edge_kmers: k-mers merged into each edge; edge_src/edge_dst: ids into node_key."""

    def __init__(self, cptr):
        c = cptr.contents
        ne, nn, nw = c.n_edges, c.n_nodes, c.key_words
        self.k, self.n_nodes, self.n_edges, self.read_bytes, self.key_words = c.k, nn, ne, c.read_bytes, nw

        def arr(ptr, n, dtype):
            return np.ctypeslib.as_array(ptr, (n,)).copy() if n else np.zeros(0, dtype)

        self.edge_src, self.edge_dst = arr(c.edge_src, ne, np.uint64), arr(c.edge_dst, ne, np.uint64)
        self.edge_weight, self.edge_kmers = arr(c.edge_weight, ne, np.uint32), arr(c.edge_kmers, ne, np.uint32)
        self.edge_label_off = np.ctypeslib.as_array(c.edge_label_off, (ne + 1,)).copy()
        self.edge_label = arr(c.edge_label, c.label_bytes, np.uint8)
        self.node_key = arr(c.node_key, nn * nw, np.uint64).reshape(nn, nw)
        self.edge_seq = []
        for i in range(ne):
            b = self.edge_label[int(self.edge_label_off[i]):int(self.edge_label_off[i + 1])]
            bases = "".join("ACGT"[(int(x) >> sh) & 3] for x in b[1:] for sh in (6, 4, 2, 0))
            self.edge_seq.append(bases[:len(bases) - int(b[0])])

    @classmethod
    def create(cls, input_files, ft, reverse_complement, minimal_weight_threshold=0, device=0, first_seen_order=False,
               remove_dead_paths=False, n_devices=1, ranks_share_device=False):
        """Build::create, optionally remove_dead_paths, then shrink -- the start of assemble_with_graph
        (asm/basic_assembler.rs:58-65) -> (GpuContigs, number_of_read_bytes)"""
        s = make_settings(K_SIZE, ft, reverse_complement, minimal_weight_threshold, device,
                          first_seen_order=first_seen_order, remove_dead_paths=remove_dead_paths, n_devices=n_devices,
                          ranks_share_device=ranks_share_device)
        cp = C.POINTER(_lib.Contigs)()
        _check(_lib.lib().katome_shrink_files(C.byref(s), _paths(input_files), len(input_files), C.byref(cp)))
        try:
            c = cls(cp)
        finally:
            _lib.lib().katome_contigs_free(cp)
        return c, c.read_bytes

    def contigs(self):
        """sorted (sequence, weight): the parity observable"""
        return sorted(zip(self.edge_seq, (int(w) for w in self.edge_weight)))


def _int_to_kmer(v, k):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def ingest_files(input_files, ft, k):
    """Host ingest alone (check_files + record scan + ACGT filter + 2-bit packing): builder.rs:57-77,118-165"""
    s = make_settings(k, ft)
    rp = C.POINTER(_lib.Reads)()
    _check(_lib.lib().katome_ingest_files(C.byref(s), _paths(input_files), len(input_files), C.byref(rp)))
    r = rp.contents
    try:
        out = dict(n_records=r.n_records, n_reads=r.n_reads, read_bytes=r.read_bytes, packed_bytes=r.packed_bytes,
                   total_windows=r.total_windows, fixed_len=r.fixed_len,
                   packed=(np.ctypeslib.as_array(r.packed, (max(r.packed_bytes, 1),)).copy()[:r.packed_bytes]),
                   byte_off=np.ctypeslib.as_array(r.byte_off, (r.n_reads + 1,)).copy(),
                   len=(np.ctypeslib.as_array(r.len, (max(r.n_reads, 1),)).copy()[:r.n_reads]))
    finally:
        _lib.lib().katome_reads_free(rp)
    return out
