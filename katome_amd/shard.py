"""The sharded (multi-GPU) build from Python: thin bindings of katome_comm_* / katome_dist_* (include/katome_gpu.h).

Everything that moves or computes is in libkatome_gpu.so (katome_amd/csrc/comm.cpp, dist.hip): this module only creates
the communicator of a one-process-per-GPU job -- RCCL, the unique id handed round with torch.distributed; or, for
rehearsals and CPU tests, a transport whose bytes travel through torch.distributed/gloo -- and wraps the handles.
A single-process caller (the Rust shim of INTEGRATION.md) needs none of this: it sets `settings.n_devices`.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from .build import KatomePanic, make_settings
from .device import Builder, _ViewOwner, _ptr, _stream, _view


def _check(status):
    if status != 0:
        raise KatomePanic(status, _lib.last_error())


def shard_range(total_reads, world, rank):
    """contiguous shard of reads for `rank`: (first, count); starts are multiples of 64 reads"""
    first, count = C.c_uint64(), C.c_uint64()
    _lib.lib().katome_shard_range(total_reads, world, rank, C.byref(first), C.byref(count))
    return first.value, count.value


class Comm:
    """one rank's end of the exchange layer"""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep                      # callbacks must outlive the handle

    @classmethod
    def rccl(cls, rank, world, device, group=None):
        """RCCL over xGMI, one process per GPU: rank 0 makes the unique id, torch.distributed hands it round"""
        L = _lib.lib()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _check(L.katome_comm_unique_id(ident))
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_uint8 * 128)(*box[0])
        h = C.c_void_p()
        _check(L.katome_comm_create_rccl(ident, rank, world, device, C.byref(h)))
        return cls(h)

    @classmethod
    def over_torch(cls, group=None, device=0):
        """the caller's transport: bytes travel through torch.distributed (gloo: staged through the host).  Lets several
        processes share one GPU (RCCL refuses that) and runs the comm layer on CPU-only machines."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def tensor_of(ptr, nbytes, on_device):
            if nbytes == 0:
                return torch.empty(0, dtype=torch.uint8)
            if on_device:
                return _view(ptr, (nbytes,), "|u1", None, torch.device("cuda", device))
            return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

        def alltoallv(_user, send, send_off, send_cnt, recv, recv_off, recv_cnt, elem, on_device):
            try:
                s_n = [send_cnt[p] * elem for p in range(world)]
                r_n = [recv_cnt[p] * elem for p in range(world)]
                s_o = [send_off[p] * elem for p in range(world)]
                r_o = [recv_off[p] * elem for p in range(world)]
                s_end, r_end = max(o + n for o, n in zip(s_o, s_n)), max(o + n for o, n in zip(r_o, r_n))
                src = tensor_of(send, s_end, on_device)
                out = tensor_of(recv, r_end, on_device)
                staged = torch.cat([src[o:o + n].cpu() for o, n in zip(s_o, s_n)]) if sum(s_n) else torch.empty(0, dtype=torch.uint8)
                got = torch.empty(sum(r_n), dtype=torch.uint8)
                dist.all_to_all_single(got, staged, output_split_sizes=r_n, input_split_sizes=s_n, group=group)
                at = 0
                for o, n in zip(r_o, r_n):
                    if n:
                        out[o:o + n].copy_(got[at:at + n])
                    at += n
                if on_device:
                    torch.cuda.synchronize()
                return 0
            except Exception as e:     # noqa: BLE001  (nothing may unwind into C)
                print("[katome comm] alltoallv callback failed: %r" % (e,), flush=True)
                return 1

        def allreduce(_user, vals, n, op):
            try:
                t = torch.tensor([vals[i] for i in range(n)], dtype=torch.int64) if n <= 64 else \
                    torch.frombuffer((C.c_int64 * n).from_address(C.addressof(vals.contents)), dtype=torch.int64).clone()
                # u64 values above 2^63 do not occur (counts, histograms, sequence numbers)
                dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[op], group=group)
                out = t.tolist()
                for i in range(n):
                    vals[i] = out[i]
                return 0
            except Exception as e:     # noqa: BLE001
                print("[katome comm] allreduce callback failed: %r" % (e,), flush=True)
                return 1

        cb = _lib.CommCallbacks(None, _lib.A2A_FN(alltoallv), _lib.ALLREDUCE_FN(allreduce))
        h = C.c_void_p()
        _check(_lib.lib().katome_comm_create_callbacks(C.byref(cb), rank, world, device, C.byref(h)))
        return cls(h, keep=cb)

    @property
    def rank(self):
        return _lib.lib().katome_comm_rank(self._h)

    @property
    def world(self):
        return _lib.lib().katome_comm_world(self._h)

    @property
    def kind(self):
        return _lib.lib().katome_comm_kind(self._h).decode()

    def set_max_message_bytes(self, n):
        _check(_lib.lib().katome_comm_set_max_message_bytes(self._h, n))

    def allreduce(self, values, op="sum"):
        arr = (C.c_uint64 * len(values))(*values)
        _check(_lib.lib().katome_comm_allreduce_u64(self._h, arr, len(values), {"sum": 0, "max": 1, "min": 2}[op]))
        return list(arr)

    def exchange_host(self, send, send_counts, elem_words=1):
        """variable all-to-all of host int64 records (elem_words words each) grouped by destination -> (recv, recv_counts)"""
        send = send.contiguous()
        world = self.world
        sc = (C.c_uint64 * world)(*send_counts)
        rc = (C.c_uint64 * world)()
        cap = 1 << 22
        recv = torch.empty(cap * elem_words, dtype=torch.int64)
        _check(_lib.lib().katome_comm_exchange(self._h, C.c_void_p(send.data_ptr()), sc, C.c_void_p(recv.data_ptr()), cap, rc,
                                               8 * elem_words, 0, None))
        counts = [int(x) for x in rc]
        return recv[:sum(counts) * elem_words], counts

    def close(self):
        if self._h:
            _lib.lib().katome_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RankGraph:
    """this rank's share of the graph (zero-copy views of the builder's device arrays): its edges, disjoint from every
    other rank's, with GLOBAL node ids; the nodes it owns.  By packed key node i has id node_base + i; in the reference's
    numbering edge_id / node_id hold the petgraph index of every edge / node."""

    def __init__(self, g, owner, device):
        ne, nn, nw = g.n_edges, g.n_nodes, g.key_words
        self.n_edges, self.n_nodes, self.total_edges, self.total_nodes = ne, nn, g.total_edges, g.total_nodes
        self.node_base, self.key_words, self.label_stride = g.node_base, nw, g.label_stride
        self.edge_key = _view(g.d_edge_key, (ne, nw), "<i8", owner, device)
        self.edge_weight = _view(g.d_edge_weight, (ne,), "<i4", owner, device)
        self.edge_src = _view(g.d_edge_src, (ne,), "<i8", owner, device)
        self.edge_dst = _view(g.d_edge_dst, (ne,), "<i8", owner, device)
        self.edge_label = _view(g.d_edge_label, (ne, g.label_stride), "|u1", owner, device)
        self.node_key = _view(g.d_node_key, (nn, nw), "<i8", owner, device)
        self.edge_id = _view(g.d_edge_id, (ne,), "<i8", owner, device) if g.d_edge_id else None
        self.node_id = _view(g.d_node_id, (nn,), "<i8", owner, device) if g.d_node_id else None
        # after remove_dead_paths on the sharded graph: the index each edge had when it was built (petgraph's adjacency order)
        self.edge_age = _view(g.d_edge_age, (ne,), "<i8", owner, device) if g.d_edge_age else None


class _InnerBuilder(Builder):
    """the single-GPU builder underneath a ShardedBuilder (owned by it): profile counters, and -- on the root after
    gather() -- the stages of device.Builder"""

    def __init__(self, handle, k, rc, device, parent=None):     # noqa: super().__init__ would create a builder
        self.parent = parent                       # the memory behind this builder's views is the parent's to free
        self.k, self.rc, self.device = k, bool(rc), device
        self.nw = _lib.lib().katome_record_words(k)
        self._h = C.c_void_p(handle)
        self.tdev = torch.device("cuda", device)

    def _retain_view(self):
        if self.parent is not None:
            self.parent._retain_view()

    def _release_view(self):
        if self.parent is not None:
            self.parent._release_view()

    def close(self):
        self._h = C.c_void_p()                     # not ours to destroy

    def _destroy(self):
        pass


class ShardedBuilder(_ViewOwner):
    """one rank's share of a sharded build (katome_dist_*); every method is collective"""

    def __init__(self, comm, k, reverse_complement, device=0, table_slots_hint=0, first_seen_order=False):
        self.comm, self.k, self.rc, self.device = comm, k, bool(reverse_complement), device
        self.first_seen = first_seen_order
        s = make_settings(k, reverse_complement=reverse_complement, device=device, table_slots_hint=table_slots_hint,
                          first_seen_order=first_seen_order)
        self._h = C.c_void_p()
        _check(_lib.lib().katome_dist_create(C.byref(s), comm._h, C.byref(self._h)))
        self.tdev = torch.device("cuda", device)

    @property
    def inner(self):
        """the rank's single-GPU builder (a fresh handle wrapper per access: no reference cycle with this object)"""
        return _InnerBuilder(_lib.lib().katome_dist_inner(self._h), self.k, self.rc, self.device, parent=self)

    @property
    def route(self):
        """which records travel: "supermers" / "local" / "tiles" (katome_dist_route)"""
        return _lib.lib().katome_dist_route(self._h).decode()

    def add_reads(self, packed, first_read, n_reads, read_len, skip=None, batch_reads=0):
        _check(_lib.lib().katome_dist_add_reads(self._h, _ptr(packed), first_read, n_reads, read_len, _ptr(skip), batch_reads, _stream()))

    def remove_weak_edges(self, threshold):
        _check(_lib.lib().katome_dist_remove_weak_edges(self._h, threshold))

    def finalize(self):
        g = _lib.DistGraph()
        _check(_lib.lib().katome_dist_finalize(self._h, C.byref(g), _stream()))
        return RankGraph(g, self, self.tdev)

    def remove_dead_paths(self):
        """Prunable::remove_dead_paths (pruner.rs:36-82) on the sharded graph, no gather (katome_dist_remove_dead_paths)
        -> (this rank's share of the pruned graph, stats dict)"""
        g, st = _lib.DistGraph(), _lib.PruneStats()
        _check(_lib.lib().katome_dist_remove_dead_paths(self._h, C.byref(g), C.byref(st), _stream()))
        return RankGraph(g, self, self.tdev), {f: getattr(st, f) for f, _ in _lib.PruneStats._fields_}

    def gather(self, root=0):
        """FIRST_SEEN_ORDER: the whole graph to `root` in the reference's index order -> on the root a builder on which
        remove_dead_paths / remove_weak_edges / standardize_* / shrink / graph() work (None elsewhere)"""
        h = C.c_void_p()
        _check(_lib.lib().katome_dist_gather(self._h, root, C.byref(h), _stream()))
        return _InnerBuilder(h.value, self.k, self.rc, self.device, parent=self) if h.value else None

    def exchange_stats(self):
        """{phase: dict(calls, bytes_out, max_message_bytes, ms)} since the last read"""
        L = _lib.lib()
        n = L.katome_dist_exchange_count()
        out = (C.c_uint64 * (4 * n))()
        _check(L.katome_dist_exchange_read(self._h, out))
        return {L.katome_dist_exchange_name(i).decode(): dict(calls=out[4 * i], bytes_out=out[4 * i + 1],
                                                             max_message_bytes=out[4 * i + 2], ms=out[4 * i + 3] / 1000.0)
                for i in range(n) if out[4 * i]}

    def _destroy(self):
        if self._h:
            _lib.lib().katome_dist_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass


class DistBuild:
    """bench.py's N > 1 job: the synthetic workload sharded over the ranks, resident in HBM before the timed region"""

    def __init__(self, wl, comm, batch_reads=0, timer=None, first_seen_order=False, min_weight=0, table_factor=2.2,
                 prune=False):
        from . import device as kd
        self.wl, self.comm, self.batch_reads, self.timer = wl, comm, batch_reads, timer
        self.first_seen, self.min_weight, self.table_factor, self.prune = first_seen_order, min_weight, table_factor, prune
        self.dev = torch.cuda.current_device()
        self.first, self.n_local = shard_range(wl.reads, comm.world, comm.rank)
        self.packed, skip = kd.synth_reads(self.first, self.n_local, wl.read_len, wl.genome_len, wl.err_rate,
                                           wl.n_inject_percent, device=self.dev)
        self.skip = skip if wl.n_inject_percent else None
        local_ok = self.n_local - (int(skip[:self.n_local].sum().item()) if wl.n_inject_percent else 0)
        self.accepted_total = comm.allreduce([local_ok], "sum")[0]
        self.exchange = {}

    def build(self):
        wl = self.wl
        hint = int(wl.expected_distinct_canonical() * self.table_factor) if self.table_factor else 0
        b = ShardedBuilder(self.comm, wl.k, wl.reverse_complement, self.dev, table_slots_hint=hint, first_seen_order=self.first_seen)
        try:
            b.inner.profile(self.timer is not None)
            if self.min_weight:
                b.remove_weak_edges(self.min_weight)
            b.add_reads(self.packed, self.first, self.n_local, wl.read_len, self.skip, self.batch_reads)
            self.route = b.route
            g = b.finalize()
            n_edges, n_nodes = g.total_edges, g.total_nodes
            if self.prune:                                  # BASELINE config 5: the pruner pass, on the sharded graph
                pg, _ = b.remove_dead_paths()
                n_edges, n_nodes = pg.total_edges, pg.total_nodes
            if self.timer is not None:
                self.timer.add(b.inner.profile_read())
                for name, x in b.exchange_stats().items():
                    acc = self.exchange.setdefault(name, dict(calls=0, bytes_out=0, max_message_bytes=0, ms=0.0))
                    acc["calls"] += x["calls"]; acc["bytes_out"] += x["bytes_out"]; acc["ms"] += x["ms"]
                    acc["max_message_bytes"] = max(acc["max_message_bytes"], x["max_message_bytes"])
            return n_edges, n_nodes
        finally:
            b.close()
