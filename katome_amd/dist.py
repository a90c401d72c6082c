"""Multi-GPU build: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference's `Build::create` (builder.rs:42-54) is a single sequential loop; what shards is the
read set (reads are independent until their k-mers meet in the graph), with ONE real exchange
step per batch:

  1. each rank extracts the k-mer records of its own reads (contiguous shard by read index);
  2. records are routed to `owner = mulhi(mix(canonical middle (k-2)-mer of the k-mer), world)` -- a single
     all-to-all of fixed-size records (RCCL has no alltoallv, so split sizes travel first in a tiny all-to-all);
  3. each rank inserts what it received into its own table: every distinct k-mer lives on exactly
     one rank, so the union of the ranks' edge lists IS the edge multiset of the whole input and
     does not depend on the number of ranks;
  4. node numbering: a k-mer shares its middle with its reverse complement and with the tail of its source
     node, so with nodes owned by `mulhi(mix(canonical last k-2 bases), world)` every out-edge of a node --
     whichever strand produced it -- already sits on the node's owner (the HmGIR shape, hm_gir.rs:91-153).
     Source ids are therefore read off the rank's own sorted edges; only the TARGET of each edge is asked
     from its owner (one key out, one id back), which is also how nodes without out-edges get registered
     (they count in node_count, stats/collections.rs:196).  Global id = rank offset + position in the
     owner's list (its nodes with out-edges ascending, then its other nodes ascending).

The 8 GPUs of an MI355X node are a full xGMI mesh, so the all-to-all runs on all 7 links of every
GPU at once and is bound by the most loaded link; balanced hashing keeps the links even.

`ops` is the set of device primitives; the product default is the HIP library (no CPU fallback).
"""
import torch
import torch.distributed as dist


class HipOps:
    """device primitives of libkatome_gpu.so (katome_amd/device.py)"""

    def __init__(self, k, rc, device, table_slots_hint=0, min_weight=0):
        """min_weight: Clean::remove_weak_edges (pruner.rs:84-93) when the edges are read out -- every k-mer's weight is
        complete at its owner by then; nodes left without edges never get an id (finalize_distributed numbers the
        endpoints of the edges that stay)"""
        from . import device as kd
        self.kd = kd
        self.k, self.rc = k, rc
        self.nw = kd.record_words(k)
        self.dev = device
        self.b = kd.Builder(k, rc, device=device, table_slots_hint=table_slots_hint)
        if min_weight:
            self.b.remove_weak_edges(min_weight)
        self.tdev = self.b.tdev

    def extract_fixed(self, packed, n_reads, read_len, skip, out, first_read):
        return self.b.extract_fixed(packed, n_reads, read_len, skip, out=out, first_read=first_read)

    def partition(self, records, n_parts, key_words=None, values=None, core=None):
        if (key_words or self.nw) == 3:
            return self._partition_three_words(records, n_parts)
        return self.b.partition(records, n_parts, key_words=key_words, values=values, core=core)

    def _partition_three_words(self, records, n_parts):
        """Tiles of 64..95 bases: the partition kernels take one- and two-word records, so the records are ordered through
        an 8-bit radix pass over (owner, position) pairs and then moved.  Any function of the tile will do as its owner --
        identical tiles only have to meet on one rank."""
        rec = records.view(-1, 3)
        n = rec.shape[0]
        if n == 0:
            return records, [0] * n_parts
        mixed = rec[:, 0] * -7046029254386353131 + (rec[:, 1] ^ (rec[:, 1] >> 29)) * -4658895280553007687 + (rec[:, 2] ^ (rec[:, 2] >> 31))
        owner = ((mixed >> 17) & 0x7FFFFFFF) % n_parts
        owner = torch.where(rec[:, 0] == -1, torch.full_like(owner, n_parts), owner)          # windows of skipped reads go last
        pos = torch.arange(n, dtype=torch.int32, device=rec.device)
        keys = owner.contiguous()
        self.kd.sort_keys(keys, 8, 1, pos, device=self.dev)                                     # stable: one 8-bit pass
        counts = [int(c) for c in torch.bincount(keys, minlength=n_parts + 1)[:n_parts].tolist()]
        return rec[pos[:sum(counts)].to(torch.int64)].reshape(-1), counts

    def insert(self, records, weights=None):
        self.b.insert(records, weights)

    # tiled counting (table.hip): (k+span-1)-mers covering `span` consecutive windows
    def tile_span(self, read_len):
        return self.b.tile_span(read_len)

    def tile_words(self, span):
        return self.b.tile_words(span)

    def tile_plan(self, read_len):
        return self.b.tile_plan(read_len)

    def extract_remainder(self, packed, n_reads, read_len, span, skip, first_read):
        return self.b.extract_remainder(packed, n_reads, read_len, span, skip, first_read=first_read)

    def extract_tiles(self, packed, n_reads, read_len, span, skip, out, first_read):
        return self.b.extract_tiles(packed, n_reads, read_len, span, skip, out=out, first_read=first_read)

    def insert_tiles(self, records, span):
        self.b.insert_tiles(records, span)

    def expand_tiles(self):
        return self.b.expand_tiles()

    def edges(self):
        return self.b.edges()

    def source_ids(self, keys):
        return self.kd.source_ids(keys.reshape(-1), self.k, self.dev)

    def target_keys(self, keys):
        return self.kd.endpoints(keys.reshape(-1), self.k, self.dev, want_src=False)[1]

    def sort_unique(self, keys, bits):
        self.kd.sort_keys(keys, bits, self.nw, device=self.dev)
        return self.kd.unique_sorted(keys, self.nw, device=self.dev)

    def rank(self, sorted_keys, queries, bits):
        return self.kd.rank_in_sorted(sorted_keys, queries, bits, self.nw, device=self.dev)

    def labels(self, keys):
        return self.kd.labels(keys.reshape(-1), self.k, self.dev)

    def empty(self, n, dtype=torch.int64):
        return torch.empty(n, dtype=dtype, device=self.tdev)

    def close(self):
        self.b.close()


class _Phases:
    """per-phase timing of the driver's own steps (HIP events on the current stream; no-op on CPU)"""

    def __init__(self, enabled):
        self.enabled = enabled and torch.cuda.is_available()
        self.pairs = []

    def __call__(self, name):
        ph = self

        class _Ctx:
            def __enter__(self):
                if ph.enabled:
                    self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    self.a.record()

            def __exit__(self, *exc):
                if ph.enabled:
                    self.b.record()
                    ph.pairs.append((name, self.a, self.b))
        return _Ctx()

    def read(self):
        if not self.enabled:
            return {}
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.pairs:
            ms, n = out.get(name, (0.0, 0))
            out[name] = (ms + a.elapsed_time(b), n + 1)
        self.pairs = []
        return out


_NO_PHASES = _Phases(False)


def _empty(n, dtype, device):
    """torch.empty that, when HBM is short, first hands the library's cached blocks back to the driver"""
    try:
        return torch.empty(n, dtype=dtype, device=device)
    except torch.OutOfMemoryError:
        if device.type == "cuda":
            from . import device as kd
            kd.release_cache(device.index if device.index is not None else torch.cuda.current_device())
            torch.cuda.empty_cache()
        return torch.empty(n, dtype=dtype, device=device)


# A single send of more than 2 GiB came back corrupted from RCCL 2.26 (measured: a 3.8 GB self-exchange);
# bigger exchanges are therefore cut into rounds of at most this many bytes per (source, destination) pair.
MAX_MESSAGE_BYTES = 1 << 30


def _a2a(out, inp, out_splits=None, in_splits=None, group=None):
    """all_to_all_single; device tensors are staged through the host when the backend cannot move them itself
    (gloo: used to rehearse several ranks on one GPU -- RCCL refuses two ranks on the same device)"""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def _all_reduce(t, op, group):
    if t.is_cuda and dist.get_backend(group) == "gloo":
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)


def _exchange(send, send_counts, nw, group):
    """all-to-all of `send` (records of nw elements, grouped by destination rank, send_counts records each).
    Returns (recv, recv_counts), recv grouped by source rank."""
    world = dist.get_world_size(group)
    sc = torch.tensor(send_counts, dtype=torch.int64, device=send.device)
    rc = torch.empty(world, dtype=torch.int64, device=send.device)
    _a2a(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.tolist()]
    recv = _empty(sum(recv_counts) * nw, send.dtype, send.device)
    chunk = max(1, MAX_MESSAGE_BYTES // (nw * send.element_size()))
    biggest = torch.tensor([max(send_counts + recv_counts + [0])], dtype=torch.int64, device=send.device)
    _all_reduce(biggest, dist.ReduceOp.MAX, group)
    rounds = (int(biggest.item()) + chunk - 1) // chunk
    if rounds <= 1:
        _a2a(recv, send[:sum(send_counts) * nw].contiguous(), [c * nw for c in recv_counts], [c * nw for c in send_counts], group)
        return recv, recv_counts
    s_off = [0] * world
    r_off = [0] * world
    for p in range(1, world):
        s_off[p] = s_off[p - 1] + send_counts[p - 1]
        r_off[p] = r_off[p - 1] + recv_counts[p - 1]
    for r in range(rounds):
        s_n = [min(max(c - r * chunk, 0), chunk) for c in send_counts]
        r_n = [min(max(c - r * chunk, 0), chunk) for c in recv_counts]
        src = torch.cat([send[(s_off[p] + r * chunk) * nw:(s_off[p] + r * chunk + s_n[p]) * nw] for p in range(world)])
        dst = _empty(sum(r_n) * nw, send.dtype, send.device)
        _a2a(dst, src, [c * nw for c in r_n], [c * nw for c in s_n], group)
        o = 0
        for p in range(world):
            recv[(r_off[p] + r * chunk) * nw:(r_off[p] + r * chunk + r_n[p]) * nw] = dst[o:o + r_n[p] * nw]
            o += r_n[p] * nw
    return recv, recv_counts


class _AsyncExchange:
    """One all-to-all of records (and, aligned with them, of values) started now and waited for later, so that the links
    work while this rank extracts / routes the next batch and counts the previous one.  Falls back to the blocking
    `_exchange` when a message would exceed MAX_MESSAGE_BYTES (rounds) or the backend cannot move device tensors (gloo)."""

    def __init__(self, parts, send_counts, group):
        """parts: [(tensor grouped by destination, elements per record)], all with the same send_counts"""
        world = dist.get_world_size(group)
        self.group, self.parts, self.send_counts = group, parts, send_counts
        dev = parts[0][0].device
        sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
        rc = torch.empty(world, dtype=torch.int64, device=dev)
        _a2a(rc, sc, group=group)
        self.recv_counts = [int(x) for x in rc.tolist()]
        biggest = max(c * nw * t.element_size() for t, nw in parts for c in send_counts + self.recv_counts) if parts else 0
        flag = torch.tensor([1 if biggest > MAX_MESSAGE_BYTES else 0], dtype=torch.int64, device=dev)
        _all_reduce(flag, dist.ReduceOp.MAX, group)                 # every rank must take the same route
        staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"
        self.work, self.recv = [], []
        if int(flag.item()) or staged:
            self.recv = [_exchange(t, send_counts, nw, group)[0] for t, nw in parts]
            return
        for t, nw in parts:
            out = _empty(sum(self.recv_counts) * nw, t.dtype, dev)
            inp = t[:sum(send_counts) * nw].contiguous()
            w = dist.all_to_all_single(out, inp, output_split_sizes=[c * nw for c in self.recv_counts],
                                       input_split_sizes=[c * nw for c in send_counts], group=group, async_op=True)
            self.work.append((w, inp))                                # the input stays alive until the wait
            self.recv.append(out)

    def wait(self):
        for w, _ in self.work:
            w.wait()
        self.work = []
        return self.recv


class RankGraph:
    """this rank's share of the graph: its edges (disjoint from every other rank's) with GLOBAL node ids,
    and the nodes it owns (global id = node_base + position)"""

    def __init__(self, edge_key, edge_weight, edge_src, edge_dst, edge_label, node_key, node_base, total_nodes,
                 total_edges):
        self.edge_key, self.edge_weight = edge_key, edge_weight
        self.edge_src, self.edge_dst, self.edge_label = edge_src, edge_dst, edge_label
        self.node_key, self.node_base = node_key, node_base
        self.n_edges = edge_weight.numel()
        self.n_nodes = node_key.shape[0]
        self.total_nodes, self.total_edges = total_nodes, total_edges


def build_shard(ops, packed, skip, n_reads, read_len, batch_reads, group=None, phases=_NO_PHASES):
    """Steps 1-3 for this rank's reads; afterwards ops' table holds the k-mers this rank owns.

    When a tile span divides the windows per read, what travels and is counted first are TILES (the
    (k+span-1)-mers covering `span` consecutive windows: span x fewer records on the links and span x fewer
    atomics); each rank then turns its distinct tiles into (k-mer, weight) records and a second, much
    smaller all-to-all brings those to the k-mers' owners."""
    world = dist.get_world_size(group)
    W = read_len - ops.k + 1
    # the plan is the same on every rank (a function of k and the read length): `span` windows per tile, `per_read` tiles
    # from the front of every read, `rest` windows left over (those travel as plain k-mer records)
    span, per_read, rest = ops.tile_plan(read_len)
    if span <= 1:
        span, per_read, rest = 1, W, 0
    nwr = ops.tile_words(span) if span > 1 else ops.nw
    kmer_core = (2, ops.k - 2)                # owner of a k-mer: its canonical middle (see the module docstring)
    recbuf = ops.empty(max(1, min(batch_reads, max(n_reads, 1)) * per_read * nwr))
    n_batches = (n_reads + batch_reads - 1) // batch_reads
    # every rank must take part in every all-to-all: agree on the number of rounds
    nb = torch.tensor([n_batches], dtype=torch.int64, device=recbuf.device)
    _all_reduce(nb, dist.ReduceOp.MAX, group)
    def count(recv):
        if recv.numel():
            if span > 1:
                ops.insert_tiles(recv, span)
            else:
                ops.insert(recv)

    # batch i travels while batch i+1 is extracted and routed and batch i-1 is counted
    in_flight = None
    for i in range(int(nb.item())):
        r0 = i * batch_reads
        nr = max(0, min(batch_reads, n_reads - r0))
        if nr:
            if span > 1:
                rec = ops.extract_tiles(packed, nr, read_len, span, skip, recbuf, r0)
            else:
                rec = ops.extract_fixed(packed, nr, read_len, skip, recbuf, r0)
            with phases("route_records"):
                part, counts = ops.partition(rec, world, key_words=nwr, core=None if span > 1 else kmer_core)
        else:
            part, counts = recbuf[:0], [0] * world
        with phases("exchange_records"):
            started = _AsyncExchange([(part, nwr)], counts, group)
            if in_flight is not None:
                count(in_flight.wait()[0])
            in_flight = started
        if rest:                                          # the windows after the last whole tile of every read
            if nr:
                rem = ops.extract_remainder(packed, nr, read_len, span, skip, r0)
                with phases("route_records"):
                    rpart, rcounts = ops.partition(rem, world, key_words=ops.nw, core=kmer_core)
            else:
                rpart, rcounts = recbuf[:0], [0] * world
            with phases("exchange_records"):
                got = _AsyncExchange([(rpart, ops.nw)], rcounts, group).wait()[0]
            if got.numel():
                ops.insert(got)
    if in_flight is not None:
        with phases("exchange_records"):
            last = in_flight.wait()[0]
        count(last)
    if span > 1:
        keys, weights = ops.expand_tiles()               # this rank's distinct tiles as (k-mer, weight) records
        # the same pipeline over slices of the record list (a slice never exceeds one message's size limit)
        n_rec = weights.numel()
        per_slice = max(1, MAX_MESSAGE_BYTES // (8 * ops.nw))
        ns = torch.tensor([(n_rec + per_slice - 1) // per_slice], dtype=torch.int64, device=recbuf.device)
        _all_reduce(ns, dist.ReduceOp.MAX, group)
        in_flight = None
        for j in range(int(ns.item())):
            a, b = min(n_rec, j * per_slice), min(n_rec, (j + 1) * per_slice)
            with phases("route_kmers"):
                if b > a:
                    pk, counts, pw = ops.partition(keys[a * ops.nw:b * ops.nw], world, key_words=ops.nw, values=weights[a:b],
                                                   core=kmer_core)
                else:
                    pk, counts, pw = keys[:0], [0] * world, weights[:0]
            with phases("exchange_kmers"):
                started = _AsyncExchange([(pk, ops.nw), (pw, 1)], counts, group)
                if in_flight is not None:
                    rk, rw = in_flight.wait()
                    if rw.numel():
                        ops.insert(rk, rw)
                in_flight = started
        if in_flight is not None:
            with phases("exchange_kmers"):
                rk, rw = in_flight.wait()
            if rw.numel():
                ops.insert(rk, rw)


def finalize_distributed(ops, group=None, phases=_NO_PHASES):
    """Step 4: sorted distinct edges of this rank + global node ids for their endpoints."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nw, k = ops.nw, ops.k
    node_bits = 2 * (k - 1)
    node_core = (0, k - 2)                            # owner of a node: its canonical tail = the middle of its out-edges
    keys, weights = ops.edges()                       # [E, nw] ascending, [E]
    E = weights.numel()
    dev = weights.device
    # every edge asks the owner of its target for the id, remembering which edge asked; while the questions travel, this
    # rank's own nodes with out-edges are read off its sorted edges (they are the sources of its own edges: no sort)
    with phases("route_targets"):
        if E:
            T = ops.target_keys(keys)
            P, counts, origin = ops.partition(T, world, key_words=nw, values=torch.arange(E, dtype=torch.int32, device=dev),
                                              core=node_core)
            del T
        else:
            P, counts, origin = ops.empty(0), [0] * world, torch.empty(0, dtype=torch.int32, device=dev)
    with phases("exchange_targets"):
        questions = _AsyncExchange([(P, nw)], counts, group)
    if E:
        with phases("local_source_ids"):
            S, lsrc = ops.source_ids(keys)
    else:
        S, lsrc = ops.empty(0), ops.empty(0)
    n_src = S.numel() // nw
    with phases("exchange_targets"):
        R = questions.wait()[0]
        recv_counts = questions.recv_counts
    del P, questions
    # answer: position among the sources, or -- for a node without out-edges -- among the (few) other nodes owned here
    with phases("answer_ids"):
        nR = R.numel() // nw
        local = ops.rank(S, R, node_bits) if (nR and n_src) else torch.full((nR,), -1, dtype=torch.int64, device=dev)
        missing = local < 0
        if nR and bool(missing.any()):
            mk = R.reshape(-1, nw)[missing].reshape(-1).contiguous()
            sinks = ops.sort_unique(mk.clone(), node_bits)
            local[missing] = ops.rank(sinks, mk, node_bits) + n_src
        else:
            sinks = ops.empty(0)
    n_owned = n_src + sinks.numel() // nw
    cdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev      # gloo gathers on the host
    pieces = [torch.empty(1, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(pieces, torch.tensor([n_owned], dtype=torch.int64, device=cdev), group=group)
    all_n = torch.cat(pieces).to(dev)
    bases = torch.cumsum(all_n, 0) - all_n
    base = int(bases[rank].item())
    total_nodes = int(all_n.sum().item())
    with phases("exchange_ids"):
        ids = local + base
        if total_nodes < (1 << 31):
            ids = ids.to(torch.int32)                                 # half the bytes on the links
        ids_P, _ = _exchange(ids, recv_counts, 1, group)              # reverse route: same split sizes, mirrored
    del R, local, ids
    if E:
        with phases("apply_ids"):
            edge_dst = _empty(E, torch.int64, dev)
            edge_dst[origin[:E].to(torch.int64)] = ids_P.to(torch.int64)   # ids_P is aligned with the routed targets
            edge_src = lsrc + base
        label = ops.labels(keys)
    else:
        edge_src = edge_dst = ops.empty(0)
        label = torch.empty((0, 1 + (k + 3) // 4), dtype=torch.uint8, device=dev)
    node_key = torch.cat([S.reshape(-1), sinks.reshape(-1)]).reshape(-1, nw)
    tot = torch.tensor([E], dtype=torch.int64, device=dev)
    _all_reduce(tot, dist.ReduceOp.SUM, group)
    return RankGraph(keys, weights, edge_src, edge_dst, label, node_key, base, total_nodes, int(tot.item()))


def shard_range(total_reads, world, rank):
    """contiguous shard of reads for `rank`; starts are multiples of 64 reads (16-byte aligned in the packed buffer)"""
    per = ((total_reads + world - 1) // world + 63) // 64 * 64
    r0 = min(total_reads, rank * per)
    r1 = min(total_reads, r0 + per)
    return r0, r1


class DistBuild:
    """bench.py's N>1 job: the synthetic workload sharded over the ranks, resident in HBM"""

    def __init__(self, wl, batch_reads, timer=None, group=None, min_weight=0):
        from . import device as kd
        self.min_weight = min_weight
        # tile records are small (8-16 tiles of 16 B per read): larger batches mean fewer exchange rounds
        self.wl, self.batch_reads, self.timer, self.group = wl, max(batch_reads, 16 * 1024 * 1024), timer, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dev = torch.cuda.current_device()
        r0, r1 = shard_range(wl.reads, self.world, self.rank)
        self.n_local = r1 - r0
        self.packed, skip = kd.synth_reads(r0, self.n_local, wl.read_len, wl.genome_len, wl.err_rate,
                                           wl.n_inject_percent, device=self.dev)
        self.skip = skip if wl.n_inject_percent else None
        acc = torch.tensor([self.n_local - (int(skip[:self.n_local].sum().item()) if wl.n_inject_percent else 0)],
                           dtype=torch.int64, device="cuda")
        _all_reduce(acc, dist.ReduceOp.SUM, group)
        self.accepted_total = int(acc.item())

    def build(self):
        wl = self.wl
        hint = int(wl.expected_distinct_canonical() * 2.2 / self.world * 1.1)
        ops = HipOps(wl.k, wl.reverse_complement, self.dev, table_slots_hint=hint, min_weight=self.min_weight)
        ops.b.profile(self.timer is not None)
        phases = _Phases(self.timer is not None)
        try:
            build_shard(ops, self.packed, self.skip, self.n_local, wl.read_len, self.batch_reads, self.group, phases)
            g = finalize_distributed(ops, self.group, phases)
            if self.timer is not None:
                self.timer.add(ops.b.profile_read())
                self.timer.add(phases.read())
            return g.n_edges, g.n_nodes
        finally:
            ops.close()
