"""Device-resident driver over the C ABI: PyTorch supplies device memory, streams and
torch.distributed (plumbing); every kernel is in katome_amd/lib/libkatome_gpu.so.

Used by bench.py (inputs resident in HBM before the timed region); the multi-GPU bindings are in katome_amd/shard.py.  Mirrors the steps of `Build::create` (reference builder.rs:142-165 ->
pt_graph.rs:277-315,172-198,333-345): extract -> insert -> finalize.
"""
import ctypes as C
import weakref

import torch

from . import _lib
from .build import KatomePanic, make_settings


def _check(status):
    if status != 0:
        raise KatomePanic(status, _lib.last_error())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class _ViewOwner:
    """owner of library memory that zero-copy views point into: close() while views are alive is put off until the last
    of them is gone (the library would hand the memory to the next build underneath them)"""

    _views = 0
    _close_pending = False

    def _retain_view(self):
        self._views += 1

    def _release_view(self):
        self._views -= 1
        if self._views == 0 and self._close_pending:
            self._close_pending = False
            self._destroy()

    def close(self):
        if self._views > 0:
            self._close_pending = True
        else:
            self._destroy()

    def _destroy(self):
        raise NotImplementedError


class _DevArray:
    """zero-copy view of library-owned device memory (kept alive by `owner`)"""

    def __init__(self, ptr, shape, typestr, owner):
        self.owner = owner
        if isinstance(owner, _ViewOwner):
            owner._retain_view()
            weakref.finalize(self, owner._release_view)
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _view(ptr, shape, typestr, owner, device):
    n = 1
    for s in shape:
        n *= s
    if n == 0 or not ptr:
        dt = {"<i8": torch.int64, "<i4": torch.int32, "|u1": torch.uint8}[typestr]
        return torch.empty(shape, dtype=dt, device=device)
    return torch.as_tensor(_DevArray(ptr, shape, typestr, owner), device=device)


def record_words(k):
    return _lib.lib().katome_record_words(k)


class DeviceGraph:
    """finalized graph, arrays resident in HBM (int64/int32 views of the u64/u32 data)"""

    def __init__(self, dg, builder, device):
        self.n_nodes, self.n_edges = dg.n_nodes, dg.n_edges
        self.key_words, self.label_stride = dg.key_words, dg.label_stride
        nw, ne, nn = dg.key_words, dg.n_edges, dg.n_nodes
        self.edge_key = _view(dg.d_edge_key, (ne, nw), "<i8", builder, device)
        self.edge_weight = _view(dg.d_edge_weight, (ne,), "<i4", builder, device)
        self.edge_src = _view(dg.d_edge_src, (ne,), "<i8", builder, device)
        self.edge_dst = _view(dg.d_edge_dst, (ne,), "<i8", builder, device)
        self.edge_label = _view(dg.d_edge_label, (ne, dg.label_stride), "|u1", builder, device)
        self.node_key = _view(dg.d_node_key, (nn, nw), "<i8", builder, device)
        # first-seen index each edge had before remove_* re-numbered it (None while age == index)
        self.edge_age = _view(dg.d_edge_age, (ne,), "<i4", builder, device) if dg.d_edge_age else None


class DeviceContigs:
    """result of Builder.shrink(): merged edges with labels of any length (compress_edge format, edge i at
    edge_label[edge_label_off[i]:edge_label_off[i+1]])"""

    def __init__(self, dc, builder, device):
        self.n_nodes, self.n_edges, self.label_bytes, self.key_words = dc.n_nodes, dc.n_edges, dc.label_bytes, dc.key_words
        ne, nn, nw = dc.n_edges, dc.n_nodes, dc.key_words
        self.edge_src = _view(dc.d_edge_src, (ne,), "<i8", builder, device)
        self.edge_dst = _view(dc.d_edge_dst, (ne,), "<i8", builder, device)
        self.edge_weight = _view(dc.d_edge_weight, (ne,), "<i4", builder, device)
        self.edge_kmers = _view(dc.d_edge_kmers, (ne,), "<i4", builder, device)
        self.edge_label_off = _view(dc.d_edge_label_off, (ne + 1,), "<i8", builder, device)
        self.edge_label = _view(dc.d_edge_label, (dc.label_bytes,), "|u1", builder, device)
        self.node_key = _view(dc.d_node_key, (nn, nw), "<i8", builder, device)

    def sequences(self):
        """host: every merged edge decoded to its ACGT string (compress.rs:283-293 decompress_edge)"""
        off = self.edge_label_off.cpu().numpy()
        lab = self.edge_label.cpu().numpy()
        out = []
        for i in range(self.n_edges):
            b = lab[off[i]:off[i + 1]]
            pad = int(b[0])
            bases = "".join("ACGT"[(int(x) >> s) & 3] for x in b[1:] for s in (6, 4, 2, 0))
            out.append(bases[:len(bases) - pad])
        return out


class Builder(_ViewOwner):
    """one GPU's share of a build"""

    def __init__(self, k, reverse_complement, device=0, table_slots_hint=0, first_seen_order=False):
        self.k, self.rc, self.device = k, bool(reverse_complement), device
        self.nw = record_words(k)
        self._settings = make_settings(k, reverse_complement=reverse_complement, device=device,
                                       table_slots_hint=table_slots_hint, first_seen_order=first_seen_order)
        self._h = C.c_void_p()
        _check(_lib.lib().katome_builder_create(C.byref(self._settings), C.byref(self._h)))
        self.tdev = torch.device("cuda", device)

    # ---- per-phase HIP-event timing (on the stream the kernels run on) ---------------------------
    def profile(self, enable=True):
        _check(_lib.lib().katome_builder_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        """{phase: (total_ms, launches, elements processed)} since the last read; synchronises.  Entries "k:<kernel>" are
        single kernels timed launch by launch inside the phases (elements: keys of a pass, slots of a scan ...)"""
        L = _lib.lib()
        n = L.katome_phase_count()
        ms, cnt, work = (C.c_double * n)(), (C.c_uint64 * n)(), (C.c_uint64 * n)()
        _check(L.katome_builder_profile_read_work(self._h, ms, cnt, work))
        return {L.katome_phase_name(i).decode(): (ms[i], int(cnt[i]), int(work[i])) for i in range(n) if cnt[i]}

    def counts(self):
        """{distinct_tiles, tile_slots, distinct_kmers, kmer_slots} of the last edges()/finalize()"""
        out = (C.c_uint64 * 8)()
        _check(_lib.lib().katome_builder_counts(self._h, out))
        return dict(distinct_tiles=out[0], tile_slots=out[1], distinct_kmers=out[2], kmer_slots=out[3],
                    distinct_mid_tiles=out[4], mid_tile_slots=out[5], span=out[6], mid_span=out[7])

    def _destroy(self):
        if self._h:
            _lib.lib().katome_builder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    # ---- extraction (compress_kmer[_with_rev_compl] over read.windows(K)) -----------------------
    def extract_fixed(self, packed, n_reads, read_len, skip=None, out=None, first_read=0):
        stride = (read_len + 3) // 4
        n_rec = n_reads * (read_len - self.k + 1) if read_len >= self.k else 0
        if out is None:
            out = torch.empty(max(n_rec, 1) * self.nw, dtype=torch.int64, device=self.tdev)
        assert out.numel() >= n_rec * self.nw
        p = C.c_void_p(packed.data_ptr() + first_read * stride)
        sk = C.c_void_p(skip.data_ptr() + first_read) if skip is not None else None
        _check(_lib.lib().katome_dev_extract_fixed(self._h, p, n_reads, read_len, sk, _ptr(out), _stream()))
        return out[:n_rec * self.nw]

    # ---- tiled counting: (k+span-1)-mers covering `span` consecutive windows ---------------------------
    def tile_span(self, read_len):
        return _lib.lib().katome_tile_span(self.k, read_len)

    def tile_words(self, span):
        return _lib.lib().katome_tile_words(self.k, span)

    def extract_tiles(self, packed, n_reads, read_len, span, skip=None, out=None, first_read=0):
        stride = (read_len + 3) // 4
        nwt = self.tile_words(span)
        n_rec = n_reads * ((read_len - self.k + 1) // span)
        if out is None:
            out = torch.empty(max(n_rec, 1) * nwt, dtype=torch.int64, device=self.tdev)
        assert out.numel() >= n_rec * nwt
        p = C.c_void_p(packed.data_ptr() + first_read * stride)
        sk = C.c_void_p(skip.data_ptr() + first_read) if skip is not None else None
        _check(_lib.lib().katome_dev_extract_tiles(self._h, p, n_reads, read_len, span, sk, _ptr(out), _stream()))
        return out[:n_rec * nwt]

    def tile_plan(self, read_len, max_tile_words=3):
        """(span, tiles per read, single windows left over per read); span 1 = count every window on its own"""
        sp, t, r = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.lib().katome_tile_plan_limited(self.k, read_len, max_tile_words, C.byref(sp), C.byref(t), C.byref(r))
        return sp.value, t.value, r.value

    def extract_remainder(self, packed, n_reads, read_len, span, skip=None, out=None, first_read=0):
        """the windows of every read that come after its last whole tile, as plain k-mer records"""
        stride = (read_len + 3) // 4
        W = read_len - self.k + 1
        n_rec = n_reads * (W % span)
        if out is None:
            out = torch.empty(max(n_rec, 1) * self.nw, dtype=torch.int64, device=self.tdev)
        assert out.numel() >= n_rec * self.nw
        p = C.c_void_p(packed.data_ptr() + first_read * stride)
        sk = C.c_void_p(skip.data_ptr() + first_read) if skip is not None else None
        _check(_lib.lib().katome_dev_extract_remainder(self._h, p, n_reads, read_len, span, sk, _ptr(out), _stream()))
        return out[:n_rec * self.nw]

    def count_reads(self, packed, n_reads, read_len, skip=None, first_read=0):
        """one batch through the library's plan: tiles (+ left-over windows), or every window on its own"""
        span, tiles, rest = self.tile_plan(read_len)
        if span > 1:
            self.count_tiles(packed, n_reads, read_len, span, skip, first_read=first_read)
            if rest:
                self.insert(self.extract_remainder(packed, n_reads, read_len, span, skip, first_read=first_read))
        else:
            self.insert(self.extract_fixed(packed, n_reads, read_len, skip, first_read=first_read))

    def count_tiles(self, packed, n_reads, read_len, span, skip=None, first_read=0):
        """extract_tiles + insert_tiles without a record buffer of the caller's (the records are made where the builder keeps them)"""
        stride = (read_len + 3) // 4
        p = C.c_void_p(packed.data_ptr() + first_read * stride)
        sk = C.c_void_p(skip.data_ptr() + first_read) if skip is not None else None
        _check(_lib.lib().katome_dev_count_tiles(self._h, p, n_reads, read_len, span, sk, _stream()))

    def insert_tiles(self, records, span):
        n = records.numel() // self.tile_words(span)
        _check(_lib.lib().katome_dev_insert_tiles(self._h, _ptr(records), n, span, _stream()))

    def expand_tiles(self):
        """(k-mer keys [n*nw] int64 view, weights [n] int32 view) of this builder's tiles; the tile table is released"""
        pk, pw, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(_lib.lib().katome_dev_expand_tiles(self._h, C.byref(pk), C.byref(pw), C.byref(n), _stream()))
        return (_view(pk.value, (n.value * self.nw,), "<i8", self, self.tdev),
                _view(pw.value, (n.value,), "<i4", self, self.tdev))

    def extract_var(self, packed, byte_off, lens, win_prefix, total_windows, out=None):
        n_reads = lens.numel()
        if out is None:
            out = torch.empty(max(total_windows, 1) * self.nw, dtype=torch.int64, device=self.tdev)
        _check(_lib.lib().katome_dev_extract_var(self._h, _ptr(packed), packed.numel(), _ptr(byte_off), _ptr(lens),
                                                 _ptr(win_prefix), n_reads, total_windows, _ptr(out), _stream()))
        return out[:total_windows * self.nw]

    # ---- routing for the multi-GPU exchange -----------------------------------------------------
    def partition(self, records, n_parts, key_words=None, values=None, core=None):
        """records grouped by owner rank (stable, invalid dropped); -> (records_out, counts[, values_out]).
        core=(shift_bits, bases): owner by the key's canonical core instead of the whole key (katome_dev_partition_core)"""
        nw = key_words or self.nw
        n = records.numel() // nw
        out = torch.empty_like(records)
        vout = torch.empty_like(values) if values is not None else None
        counts = (C.c_uint64 * n_parts)()
        if core is None:
            _check(_lib.lib().katome_dev_partition(self.device, _ptr(records), _ptr(values), n, nw, n_parts, _ptr(out), _ptr(vout),
                                                   counts, _stream()))
        else:
            _check(_lib.lib().katome_dev_partition_core(self.device, _ptr(records), _ptr(values), n, nw, core[0], core[1], n_parts,
                                                        _ptr(out), _ptr(vout), counts, _stream()))
        counts = [int(c) for c in counts]
        return (out, counts) if values is None else (out, counts, vout)

    # ---- add_single_edge_fastaq for a batch ------------------------------------------------------
    def insert(self, records, weights=None):
        n = records.numel() // self.nw
        if weights is None:
            _check(_lib.lib().katome_dev_insert(self._h, _ptr(records), n, _stream()))
        else:
            _check(_lib.lib().katome_dev_insert_weighted(self._h, _ptr(records), _ptr(weights), n, _stream()))

    def remove_weak_edges(self, threshold):
        """Clean::remove_weak_edges (pruner.rs:84-93).  Before finalize(): applied when the edges are read out (first-seen
        order: after the numbering, with petgraph's retain_edges / retain_nodes re-numbering).  On a finalized
        first-seen-order builder: applied now, same re-numbering; fetch the arrays again with graph()."""
        _check(_lib.lib().katome_dev_remove_weak_edges(self._h, threshold, _stream()))

    def standardize_contigs(self):
        """Standardizable::standardize_contigs (standardizer.rs:72-122) on the finalized graph, in place"""
        _check(_lib.lib().katome_dev_standardize_contigs(self._h, _stream()))

    def standardize_edges(self, original_genome_length, threshold):
        """Standardizable::standardize_edges (standardizer.rs:42-70): scale, round, remove_weak_edges(1)"""
        _check(_lib.lib().katome_dev_standardize_edges(self._h, original_genome_length, threshold, _stream()))

    def shrink(self, mode=None):
        """Shrinkable::shrink (shrinker.rs:165-209) of the finalized graph as it stands -> DeviceContigs.  mode "exact": the
        reference's own cuts and numbering, index for index (its traversal order on one host core, the bytes on the device);
        "fast": traversal-free, all on the device, this library's numbering; None: exact on a first-seen-order builder.
        `self.last_shrink_host_ms`: the sequential part of the last exact call"""
        dc = _lib.DevContigs()
        host_ms = C.c_double(0)
        _check(_lib.lib().katome_dev_shrink_mode(self._h, {None: 0, "auto": 0, "fast": 1, "exact": 2}[mode], C.byref(dc), C.byref(host_ms),
                                                 _stream()))
        self.last_shrink_host_ms = host_ms.value
        return DeviceContigs(dc, self, self.tdev)

    def graph(self):
        """the finalized graph as it stands (after remove_dead_paths / remove_weak_edges)"""
        dg = _lib.DevGraph()
        _check(_lib.lib().katome_dev_current_graph(self._h, C.byref(dg)))
        return DeviceGraph(dg, self, self.tdev)

    def table_count(self):
        out = C.c_uint64()
        _check(_lib.lib().katome_dev_table_count(self._h, C.byref(out), _stream()))
        return out.value

    # ---- PtGraph::create post-pass -----------------------------------------------------------------
    def edges(self):
        """sorted distinct oriented edges of this builder's table: (keys [n, nw] int64 view, weights [n] int32 view)"""
        pk, pw, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(_lib.lib().katome_dev_edges(self._h, C.byref(pk), C.byref(pw), C.byref(n), _stream()))
        return (_view(pk.value, (n.value, self.nw), "<i8", self, self.tdev),
                _view(pw.value, (n.value,), "<i4", self, self.tdev))

    def finalize(self):
        dg = _lib.DevGraph()
        _check(_lib.lib().katome_dev_finalize(self._h, C.byref(dg), _stream()))
        return DeviceGraph(dg, self, self.tdev)

    def remove_dead_paths(self):
        """Prunable::remove_dead_paths (pruner.rs:36-82) on the finalized first-seen-ordered graph, in place
        -> (DeviceGraph, stats dict)"""
        dg, st = _lib.DevGraph(), _lib.PruneStats()
        _check(_lib.lib().katome_dev_remove_dead_paths(self._h, C.byref(dg), C.byref(st), _stream()))
        return DeviceGraph(dg, self, self.tdev), {f: getattr(st, f) for f, _ in _lib.PruneStats._fields_}


# ---- primitives ------------------------------------------------------------------------------------
def sort_keys(keys, key_bits, key_words, values=None, device=0):
    n = keys.numel() // key_words
    _check(_lib.lib().katome_dev_sort(device, _ptr(keys), _ptr(values), n, key_words, key_bits, _stream()))
    return keys, values


def unique_sorted(keys, key_words, device=0):
    n = keys.numel() // key_words
    out = C.c_uint64()
    _check(_lib.lib().katome_dev_unique(device, _ptr(keys), n, key_words, C.byref(out), _stream()))
    return keys.view(-1)[:out.value * key_words]


def replay_edge_removals(pos, mult, n_edges, device=0):
    """remove_paths' swap_removes (pruner.rs:199-217) for marked positions `pos` (ascending, int32/uint32 on the device)
    listed `mult` times each -> (victims in removal order, move_to, move_from, edges left, removals owed to repeats)"""
    u = pos.numel()
    marks = int(mult.to(torch.int64).sum().item()) if u else 0
    victims = torch.empty(max(marks, 1), dtype=torch.int32, device=pos.device)
    to = torch.empty(max(u, 1), dtype=torch.int32, device=pos.device)
    frm = torch.empty(max(u, 1), dtype=torch.int32, device=pos.device)
    counts = (C.c_uint64 * 4)()
    _check(_lib.lib().katome_dev_replay_edge_removals(device, _ptr(pos), _ptr(mult), u, n_edges, _ptr(victims), _ptr(to), _ptr(frm),
                                                      C.cast(counts, C.c_void_p), _stream()))
    m, moves, left, dups = (int(x) for x in counts)
    return victims[:m], to[:moves], frm[:moves], left, dups


def replay_node_removals(die, n_nodes, device=0):
    """remove_single_node after every removed edge (pruner.rs:206-225) for die[t] = (source, target) left without edges by
    edge removal t (-1 = stays) -> (move_to, move_from, nodes left, gave_up)"""
    m = die.numel() // 2
    to = torch.empty(max(2 * m, 1), dtype=torch.int32, device=die.device)
    frm = torch.empty(max(2 * m, 1), dtype=torch.int32, device=die.device)
    counts = (C.c_uint64 * 3)()
    _check(_lib.lib().katome_dev_replay_node_removals(device, _ptr(die), m, n_nodes, _ptr(to), _ptr(frm), C.cast(counts, C.c_void_p), _stream()))
    moves, left, gave_up = (int(x) for x in counts)
    return to[:moves], frm[:moves], left, bool(gave_up)


def replay_edge_removals64(pos, mult, n_edges, device=0):
    """replay_edge_removals on 64-bit positions (int64 tensors; the positions of a graph sharded over several GPUs)"""
    u = pos.numel()
    marks = int(mult.to(torch.int64).sum().item()) if u else 0
    victims = torch.empty(max(marks, 1), dtype=torch.int64, device=pos.device)
    to = torch.empty(max(u, 1), dtype=torch.int64, device=pos.device)
    frm = torch.empty(max(u, 1), dtype=torch.int64, device=pos.device)
    counts = (C.c_uint64 * 4)()
    _check(_lib.lib().katome_dev_replay_edge_removals64(device, _ptr(pos), _ptr(mult), u, n_edges, _ptr(victims), _ptr(to), _ptr(frm),
                                                        C.cast(counts, C.c_void_p), _stream()))
    m, moves, left, dups = (int(x) for x in counts)
    return victims[:m], to[:moves], frm[:moves], left, dups


def replay_node_removals64(die, n_nodes, device=0):
    """replay_node_removals on 64-bit positions (-1 = stays)"""
    m = die.numel() // 2
    to = torch.empty(max(2 * m, 1), dtype=torch.int64, device=die.device)
    frm = torch.empty(max(2 * m, 1), dtype=torch.int64, device=die.device)
    counts = (C.c_uint64 * 3)()
    _check(_lib.lib().katome_dev_replay_node_removals64(device, _ptr(die), m, n_nodes, _ptr(to), _ptr(frm), C.cast(counts, C.c_void_p), _stream()))
    moves, left, gave_up = (int(x) for x in counts)
    return to[:moves], frm[:moves], left, bool(gave_up)


def scan_counts(counts, device=0):
    """exclusive prefix sums (int64, one more entry than counts: the total) of int32/uint32 counts on the device"""
    m = counts.numel()
    offs = torch.empty(m + 1, dtype=torch.int64, device=counts.device)
    _check(_lib.lib().katome_dev_scan_counts(device, _ptr(counts), m, _ptr(offs), _stream()))
    return offs


def rank_in_sorted(sorted_keys, queries, key_bits, key_words, device=0):
    ns, nq = sorted_keys.numel() // key_words, queries.numel() // key_words
    out = torch.empty(max(nq, 1), dtype=torch.int64, device=queries.device)
    _check(_lib.lib().katome_dev_rank(device, _ptr(sorted_keys), ns, key_words, key_bits, _ptr(queries), nq, _ptr(out),
                                      _stream()))
    return out[:nq]


def release_cache(device=0):
    """hand the library's cached device blocks back to the driver -> bytes the library still holds afterwards (what live
    builders, results and views pin; 0 when everything was closed)"""
    _check(_lib.lib().katome_dev_release_cache(device))
    return cache_stats(device)["held_bytes"]


def cache_stats(device=0):
    out = (C.c_uint64 * 3)()
    _check(_lib.lib().katome_dev_cache_stats(device, out))
    return dict(held_bytes=int(out[0]), free_bytes=int(out[1]), live_blocks=int(out[2]))


def source_ids(edge_keys, k, device=0):
    """distinct source (k-1)-mers of sorted distinct edges, ascending, and each edge's position among them
    -> (node_keys [n_src*nw], edge_src [E])"""
    nw = record_words(k)
    n = edge_keys.numel() // nw
    nodes = torch.empty(max(n, 1) * nw, dtype=torch.int64, device=edge_keys.device)
    src = torch.empty(max(n, 1), dtype=torch.int64, device=edge_keys.device)
    ns = C.c_uint64()
    _check(_lib.lib().katome_dev_source_ids(device, _ptr(edge_keys), n, k, _ptr(nodes), _ptr(src), C.byref(ns), _stream()))
    return nodes[:ns.value * nw], src[:n]


def key_owner(key_words_list, n_parts, core=None):
    """host: owner of one key given as its u64 words (most significant first)"""
    nw = len(key_words_list)
    arr = (C.c_uint64 * nw)(*key_words_list)
    return _lib.lib().katome_key_owner(arr, nw, core[0] if core else 0, core[1] if core else 0, n_parts)


def node_ids(edge_keys, k, device=0):
    """local node numbering of sorted distinct edges -> (node_keys [n_nodes*nw], edge_src [E], edge_dst [E])"""
    nw = record_words(k)
    n = edge_keys.numel() // nw
    nodes = torch.empty(max(2 * n, 1) * nw, dtype=torch.int64, device=edge_keys.device)
    src = torch.empty(max(n, 1), dtype=torch.int64, device=edge_keys.device)
    dst = torch.empty(max(n, 1), dtype=torch.int64, device=edge_keys.device)
    nn = C.c_uint64()
    _check(_lib.lib().katome_dev_node_ids(device, _ptr(edge_keys), n, k, _ptr(nodes), _ptr(src), _ptr(dst), C.byref(nn), _stream()))
    return nodes[:nn.value * nw], src[:n], dst[:n]


def endpoints(edge_keys, k, device=0, want_src=True):
    nw = record_words(k)
    n = edge_keys.numel() // nw
    src, dst = (torch.empty_like(edge_keys) if want_src else None), torch.empty_like(edge_keys)
    _check(_lib.lib().katome_dev_endpoints(device, _ptr(edge_keys), n, k, _ptr(src), _ptr(dst), _stream()))
    return src, dst


def labels(edge_keys, k, device=0):
    nw = record_words(k)
    n = edge_keys.numel() // nw
    stride = 1 + (k + 3) // 4
    out = torch.empty((max(n, 1) * stride + 3) // 4 * 4, dtype=torch.uint8, device=edge_keys.device)
    _check(_lib.lib().katome_dev_labels(device, _ptr(edge_keys), n, k, _ptr(out), _stream()))
    return out[:n * stride].view(n, stride)


def synth_reads(first_read, n_reads, read_len, genome_len, err_rate, n_inject_percent=0, device=0):
    """deterministic synthetic workload, generated in HBM (DESIGN.md 'Synthetic workload')"""
    tdev = torch.device("cuda", device)
    stride = (read_len + 3) // 4
    packed = torch.empty(n_reads * stride + 32, dtype=torch.uint8, device=tdev)
    skip = torch.empty(max(n_reads, 1), dtype=torch.uint8, device=tdev)
    _check(_lib.lib().katome_dev_synth_reads(device, first_read, n_reads, read_len, genome_len, float(err_rate),
                                             n_inject_percent, _ptr(packed), _ptr(skip), _stream()))
    return packed, skip
