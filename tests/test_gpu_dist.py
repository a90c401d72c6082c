"""The sharded (multi-GPU) build through the C ABI (`settings.n_devices`, katome_amd/csrc/dist.hip): on a one-GPU box the
ranks share the card (KATOME_FLAG_RANKS_SHARE_DEVICE: peer copies instead of RCCL, same sharding / routing / exchange /
global-numbering code), and RCCL itself runs at world size 1.  Parity against the oracle's single sequential build:
by packed key the edge multiset and node set; in the reference's numbering every array index for index, whatever the
number of ranks -- including BASELINE config 5's shape (k=63, three-word tiles, remove_dead_paths)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import int_to_kmer, pack_reads_ascii  # noqa: E402

pytestmark = pytest.mark.gpu


def _reads(oracle, n, L, genome=3000, err=2e-2, npct=4):
    ascii_reads = oracle.synth_reads(0, n, L, genome, err, npct)
    has_n = (ascii_reads == ord("N")).any(axis=1)
    clean = ascii_reads.copy()
    clean[clean == ord("N")] = ord("A")
    return ascii_reads, pack_reads_ascii(clean).reshape(-1).copy(), has_n.astype(np.uint8)


def _same_arrays(g, ref):
    assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges)
    assert np.array_equal(g.edge_label, ref.edge_label)          # same edge at every index
    assert np.array_equal(g.edge_weight, ref.edge_weight)
    assert np.array_equal(g.edge_src, ref.edge_src) and np.array_equal(g.edge_dst, ref.edge_dst)


# katome_tile_plan: L=50: k=11 -> 2 tiles of 20 windows; k=33 -> 1 tile of 18 (two-word key, one three-word tile of 50 bases
# = two words ...); L=150, k=63: 3 tiles of 28 windows (90-mers: THREE-word tiles) + 4 windows; L=53, k=11: 43 windows
# = 2 tiles of 21 + 1 left over; k=60, L=75: 16 windows -> one tile of 16 (75 bases, three words)
CASES = [(2, 11, True, 260, 50), (3, 12, False, 200, 50), (2, 33, True, 130, 50), (4, 11, True, 300, 53),
         (2, 63, True, 120, 150), (3, 60, False, 90, 75), (8, 31, True, 700, 150), (3, 11, True, 100, 53),
         (4, 40, True, 64, 100)]           # the last ones leave ranks without reads (shards are multiples of 64 reads)


@pytest.mark.parametrize("world,k,rc,n,L", CASES)
def test_sharded_build_by_packed_key_equals_oracle(oracle, world, k, rc, n, L):
    from katome_amd.build import GpuGraph
    ascii_reads, packed, skip = _reads(oracle, n, L)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=world,
                                        ranks_share_device=True)
    ref = oracle.build_ascii(ascii_reads, k, rc)
    assert rb == ref.read_bytes
    assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges)
    assert g.multiset() == ref.multiset()
    ek, nk = g.key_ints("edge"), g.key_ints("node")
    assert len(set(ek)) == len(ek) and len(set(nk)) == len(nk)       # every k-mer on exactly one rank, every node owned once
    mask = (1 << (2 * (k - 1))) - 1
    for e in range(g.n_edges):
        assert nk[int(g.edge_src[e])] == ek[e] >> 2 and nk[int(g.edge_dst[e])] == ek[e] & mask
    assert set(g.edge_src.tolist()) | set(g.edge_dst.tolist()) == set(range(g.n_nodes))
    oracle.set_k(k)
    for e in range(0, g.n_edges, max(1, g.n_edges // 200)):
        assert bytes(g.edge_label[e]) == oracle.compress_edge(int_to_kmer(ek[e], k).encode())


@pytest.mark.parametrize("world,k,rc,n,L", CASES)
def test_sharded_build_in_reference_order_equals_oracle(oracle, world, k, rc, n, L):
    """KATOME_FLAG_FIRST_SEEN_ORDER on the sharded route: petgraph's own edge and node indices (pt_graph.rs:149,194) from
    global ranks of the first-insertion sequence numbers -- identical to the sequential build for any number of ranks"""
    from katome_amd.build import GpuGraph
    ascii_reads, packed, skip = _reads(oracle, n, L)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=world,
                                        ranks_share_device=True, first_seen_order=True)
    ref = oracle.build_ascii(ascii_reads, k, rc)
    assert rb == ref.read_bytes
    _same_arrays(g, ref)


@pytest.mark.parametrize("prune_route", ["sharded", "sharded-hop", "gather"])
@pytest.mark.parametrize("world,k,rc,n,L,genome,err", [(2, 63, True, 400, 150, 4000, 5e-4), (4, 63, True, 900, 150, 9000, 3e-4),
                                                       (8, 63, False, 600, 150, 4000, 5e-4), (2, 63, True, 400, 150, 4000, 5e-3),
                                                       (3, 31, True, 1500, 150, 9000, 1e-2), (2, 40, False, 800, 100, 5000, 2e-3)])
def test_config5_shape_pruned_on_the_sharded_route(oracle, monkeypatch, world, k, rc, n, L, genome, err, prune_route):
    """BASELINE config 5's shape: k=63 (two-word keys, three-word tiles), reads sharded over the ranks, then the reference's
    first pruning -- Prunable::remove_dead_paths (pruner.rs:36-82), whose walks follow petgraph's adjacency and whose
    swap_removes re-number by index.  "sharded": on the sharded graph itself (katome_dist_remove_dead_paths: the walks run on
    every rank's copy of the successor table, the index replays on 64-bit positions; no gather, every rank writes its share
    of the result); "sharded-hop": the same with walkers that hop between the owners (what runs when the table does not fit);
    "gather": on the graph gathered to one rank (KATOME_DIST_PRUNE=gather).  Index for index against the oracle's literal
    petgraph, ages included."""
    from katome_amd.build import GpuGraph
    if prune_route == "gather":
        monkeypatch.setenv("KATOME_DIST_PRUNE", "gather")
    if prune_route == "sharded-hop":
        monkeypatch.setenv("KATOME_DIST_PRUNE_WALKS", "hop")
    ascii_reads, packed, skip = _reads(oracle, n, L, genome, err, 1)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=world,
                                        ranks_share_device=True, first_seen_order=True, remove_dead_paths=True)
    ref = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=True)
    full = oracle.build_ascii(ascii_reads, k, rc)
    assert ref.n_edges < full.n_edges and (ref.n_edges > 0 or err > 1e-3)   # something was pruned; at the high error rate: everything
    _same_arrays(g, ref)
    if ref.n_edges:
        assert g.edge_age is not None and np.array_equal(g.edge_age.astype(np.uint64) + 1, ref.edge_slot)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("where", ["edges", "nodes"])
def test_a_failure_on_rank_0_inside_the_pruning_reaches_every_rank(oracle, monkeypatch, where):
    """rank 0 alone replays the removals of a pass; when that fails (here: forced, KATOME_DIST_PRUNE_FAIL) its status travels
    with the counts all ranks agree on next, so every rank returns the error -- none is left waiting in a collective.  The
    thread ranks of an n_devices build join before katome_build_packed returns: a rank still waiting would hang this test."""
    from katome_amd.build import GpuGraph, KatomePanic
    monkeypatch.setenv("KATOME_DIST_PRUNE_FAIL", where)
    ascii_reads, packed, skip = _reads(oracle, 400, 150, 4000, 5e-4, 1)
    with pytest.raises(KatomePanic) as e:
        GpuGraph.create_from_packed(packed, 400, 150, skip=skip, reverse_complement=True, k=63, n_devices=3,
                                    ranks_share_device=True, first_seen_order=True, remove_dead_paths=True)
    assert "E_UNSUPPORTED" in str(e.value) or "distributed pruning" in str(e.value)
    monkeypatch.delenv("KATOME_DIST_PRUNE_FAIL")
    g, _ = GpuGraph.create_from_packed(packed, 400, 150, skip=skip, reverse_complement=True, k=63, n_devices=3,
                                       ranks_share_device=True, first_seen_order=True, remove_dead_paths=True)
    _same_arrays(g, oracle.build_ascii(ascii_reads, 63, True, remove_dead_paths=True))      # (and the library is fine afterwards)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("route", ["local", "tiles", "supermers"])
def test_a_rank_whose_reads_are_refused_does_not_leave_the_others_in_the_exchange(oracle, monkeypatch, route):
    """a rank that fails BEFORE the exchange (its reads do not fit, or are of a kind the route does not take: forced here with
    KATOME_DIST_ADD_FAIL) never enters finalize's collectives; the thread ranks of an n_devices build meet once between taking
    the reads and finalize, the failed rank has poisoned that meeting, and katome_build_packed returns ITS error instead of
    hanging in the others' exchange"""
    from katome_amd.build import GpuGraph, KatomePanic
    monkeypatch.setenv("KATOME_DIST_ROUTE", route)
    monkeypatch.setenv("KATOME_DIST_ADD_FAIL", "1")
    ascii_reads, packed, skip = _reads(oracle, 600, 100, 4000, 1e-3, 1)
    with pytest.raises(KatomePanic) as e:
        GpuGraph.create_from_packed(packed, 600, 100, skip=skip, reverse_complement=True, k=21, n_devices=3, ranks_share_device=True)
    assert "rank 1 of 3" in str(e.value) and "KATOME_DIST_ADD_FAIL" in str(e.value)
    monkeypatch.delenv("KATOME_DIST_ADD_FAIL")
    g, _ = GpuGraph.create_from_packed(packed, 600, 100, skip=skip, reverse_complement=True, k=21, n_devices=3, ranks_share_device=True)
    assert g.multiset() == oracle.build_ascii(ascii_reads, 21, True).multiset()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,k,rc,n,L", [(3, 16, False, 293, 123), (2, 15, True, 5000, 142), (4, 18, True, 900, 140)])
def test_supermer_route_with_more_runs_than_slots(oracle, monkeypatch, world, k, rc, n, L):
    """short minimizer windows (k = 15..18: w = 3..6) cut a long read into more runs than it has slots and more than a lane lists:
    the rest goes window by window into the call's spill region, and a region that proves too small is made the size the kernel
    counted and the reads are cut again (dist.hip katome_dist_add_reads) -- never an error on one rank alone.  (The first case
    is one a fuzz run found hanging: one rank's region overflowed, the others waited for it in the exchange)"""
    from katome_amd.build import GpuGraph
    monkeypatch.setenv("KATOME_DIST_ROUTE", "supermers")
    ascii_reads, packed, skip = _reads(oracle, n, L, 4000, 1e-3, 0)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=world, ranks_share_device=True)
    ref = oracle.build_ascii(ascii_reads, k, rc)
    assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges) and g.multiset() == ref.multiset()


@pytest.mark.parametrize("world,k,rc,n,L,genome,err", [(2, 63, True, 400, 150, 4000, 5e-4), (3, 31, True, 1500, 150, 9000, 1e-2)])
def test_sharded_pruning_with_the_node_replay_on_the_host(oracle, monkeypatch, world, k, rc, n, L, genome, err):
    """a pass whose node moves chain further than the device replay follows is replayed by the sequential statement on rank 0's
    host (64-bit positions; forced here with KATOME_PRUNE_HOST_NODES): the same arrays"""
    from katome_amd.build import GpuGraph
    monkeypatch.setenv("KATOME_PRUNE_HOST_NODES", "1")
    ascii_reads, packed, skip = _reads(oracle, n, L, genome, err, 1)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=world,
                                        ranks_share_device=True, first_seen_order=True, remove_dead_paths=True)
    ref = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=True)
    _same_arrays(g, ref)
    if ref.n_edges:
        assert np.array_equal(g.edge_age.astype(np.uint64) + 1, ref.edge_slot)


def test_sharded_pruning_called_twice_and_after_a_gather(oracle):
    """Prunable::remove_dead_paths may be called again (asm/basic_assembler.rs:59,73): on the sharded graph the second call is the
    reference's one empty pass -- the same share, no device fault --; once the shares were gathered it is an error, like a second
    gather, never a walk over released arrays"""
    from katome_amd import shard as ks
    from katome_amd.build import KatomePanic
    k, rc, n, L = 31, True, 1500, 150
    ascii_reads, packed, skip = _reads(oracle, n, L, 9000, 1e-2, 1)
    pt = torch.from_numpy(np.concatenate([packed, np.zeros(32, np.uint8)])).cuda()
    st = torch.from_numpy(np.concatenate([skip, np.zeros(16, np.uint8)])).cuda()
    comm = ks.Comm.rccl(0, 1, 0)
    b = ks.ShardedBuilder(comm, k, rc, 0, first_seen_order=True)
    try:
        b.add_reads(pt, 0, n, L, st, 0)
        g = b.finalize()
        full_edges = g.total_edges
        del g
        pg, s1 = b.remove_dead_paths()
        ref = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=True)
        assert (pg.total_nodes, pg.total_edges) == (ref.n_nodes, ref.n_edges) and ref.n_edges < full_edges
        first = (pg.edge_id.cpu().numpy().copy(), pg.edge_src.cpu().numpy().copy(), pg.edge_weight.cpu().numpy().copy())
        del pg
        pg2, s2 = b.remove_dead_paths()
        assert s2["passes"] == 1 and s2["removed_edges"] == 0 and s1["removed_edges"] == full_edges - ref.n_edges
        assert (pg2.total_nodes, pg2.total_edges) == (ref.n_nodes, ref.n_edges)
        assert np.array_equal(pg2.edge_id.cpu().numpy(), first[0]) and np.array_equal(pg2.edge_src.cpu().numpy(), first[1])
        assert np.array_equal(pg2.edge_weight.cpu().numpy(), first[2])
        del pg2
        root = b.gather(0)
        dg = root.graph()
        assert (dg.n_nodes, dg.n_edges) == (ref.n_nodes, ref.n_edges)
        assert np.array_equal(dg.edge_src.cpu().numpy().astype(np.uint64), ref.edge_src)
        del dg, root
        with pytest.raises(KatomePanic):
            b.remove_dead_paths()
        with pytest.raises(KatomePanic):
            b.gather(0)
    finally:
        b.close()
        comm.close()


def test_every_stage_after_a_sharded_build(oracle, tmp_path):
    """katome_build_files_staged with n_devices: the FASTQ fixture sharded over three ranks, every stage of
    assemble_with_graph before collapse on the gathered graph; the graph collapse() receives, index for index"""
    from katome_amd.build import GpuGraph, InputFileType, set_global_k_sizes
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data3.txt")
    set_global_k_sizes(31)
    oracle.set_genome_length(20000)
    g, rb = GpuGraph.create([path], InputFileType.Fastq, True, 2, first_seen_order=True, stages="dcwced",
                            original_genome_length=20000, n_devices=3, ranks_share_device=True)
    ref = oracle.build_files([path], 31, True, remove_weak_edges=2, stages="dcwced")
    assert rb == ref.read_bytes
    _same_arrays(g, ref)


def test_shrink_after_a_sharded_build(oracle):
    """katome_shrink_files with n_devices: build over three ranks in the reference's numbering, first pruning and shrink on
    the gathered graph -- the same merged edges as the one-GPU call (and the reference's counts on the fixture)"""
    from katome_amd.build import GpuContigs, InputFileType, set_global_k_sizes
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data3.txt")
    set_global_k_sizes(40)
    one, rb1 = GpuContigs.create([path], InputFileType.Fastq, False, 0, first_seen_order=True)
    many, rbn = GpuContigs.create([path], InputFileType.Fastq, False, 0, first_seen_order=True, n_devices=3, ranks_share_device=True)
    assert rb1 == rbn == 23300
    assert (many.n_nodes, many.n_edges) == (one.n_nodes, one.n_edges) == (466, 233)          # tests/shrinker.rs:33-36
    assert many.contigs() == one.contigs()
    assert np.array_equal(many.edge_src, one.edge_src) and np.array_equal(many.edge_dst, one.edge_dst)


def test_inputs_the_sharded_route_does_not_take_are_built_on_one_gpu(oracle, tmp_path):
    """reads of unequal length and BFCounter input: n_devices is a resource hint, the result is the same graph"""
    from katome_amd.build import GpuGraph, InputFileType, set_global_k_sizes
    rng = np.random.default_rng(3)
    lines = []
    for i in range(120):
        n = int(rng.integers(40, 90))
        lines += ["@r%d" % i, "".join("ACGT"[c] for c in rng.integers(0, 4, n)), "+", "I" * n]
    fq = tmp_path / "var.fq"
    fq.write_text("\n".join(lines) + "\n")
    set_global_k_sizes(21)
    g, rb = GpuGraph.create([str(fq)], InputFileType.Fastq, True, 0, n_devices=2, ranks_share_device=True)
    ref = oracle.build_files([str(fq)], 21, True)
    assert rb == ref.read_bytes and g.multiset() == ref.multiset()


def test_too_many_devices_is_an_error():
    from katome_amd.build import GpuGraph, KatomePanic
    packed = np.zeros(64 * 13, np.uint8)
    with pytest.raises(KatomePanic) as e:
        GpuGraph.create_from_packed(packed, 64, 50, reverse_complement=True, k=11, n_devices=torch.cuda.device_count() + 1)
    assert e.value.name == "E_DEVICE" and "visible" in e.value.message
    with pytest.raises(KatomePanic) as e:      # pruning needs the reference's numbering, on any number of GPUs
        GpuGraph.create_from_packed(packed, 64, 50, reverse_complement=True, k=11, n_devices=2, ranks_share_device=True,
                                    remove_dead_paths=True)
    assert e.value.name == "E_ARG"


@pytest.mark.parametrize("first_seen", [False, True])
def test_thread_rank_over_rccl(oracle, monkeypatch, first_seen):
    """the in-process driver with the RCCL transport (ncclCommInitAll, a rank thread, grouped send/recv to self): what
    settings.n_devices > 1 uses on a multi-GPU node, here as a world of one"""
    from katome_amd.build import GpuGraph
    monkeypatch.setenv("KATOME_FORCE_SHARDED", "1")
    k, rc, n, L = 31, True, 900, 150
    ascii_reads, packed, skip = _reads(oracle, n, L, 20000, 5e-3, 2)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=1, first_seen_order=first_seen,
                                        remove_dead_paths=first_seen)
    ref = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=first_seen)
    assert rb == ref.read_bytes
    if first_seen:
        _same_arrays(g, ref)
    else:
        assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges) and g.multiset() == ref.multiset()


def _dist_build_one_process(k, rc, packed_t, skip_t, n, L, first_seen, comm, dev=0):
    """one rank (of a world of 1) driving katome_dist_* itself, as a process-per-GPU job does"""
    from katome_amd import _lib
    from katome_amd.build import make_settings
    L_ = _lib.lib()
    s = make_settings(k, reverse_complement=rc, device=dev, first_seen_order=first_seen)
    d = C.c_void_p()
    assert L_.katome_dist_create(C.byref(s), comm, C.byref(d)) == 0, _lib.last_error()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L_.katome_dist_add_reads(d, C.c_void_p(packed_t.data_ptr()), 0, n, L, C.c_void_p(skip_t.data_ptr()), 256, stream) == 0, _lib.last_error()
    g = _lib.DistGraph()
    assert L_.katome_dist_finalize(d, C.byref(g), stream) == 0, _lib.last_error()
    return d, g


@pytest.mark.parametrize("route", ["local", "tiles"])
@pytest.mark.parametrize("world,k,rc,L", [(2, 31, True, 150), (3, 31, True, 150), (8, 31, True, 150), (2, 63, True, 150), (4, 40, False, 100)])
def test_both_routes_of_the_sharded_build(oracle, monkeypatch, route, world, k, rc, L):
    """Few ranks count their own reads and route distinct k-mers ("local"), many route tiles, mid tiles and k-mer records level by
    level ("tiles"); the library picks by world size, KATOME_DIST_ROUTE overrides.  Either way, at every world size: the
    reference's numbering, all four arrays index for index."""
    from katome_amd.build import GpuGraph
    monkeypatch.setenv("KATOME_DIST_ROUTE", route)
    n = 1500
    ascii_reads, packed, skip = _reads(oracle, n, L, 9000, 3e-3, 1)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=world,
                                        ranks_share_device=True, first_seen_order=True)
    _same_arrays(g, oracle.build_ascii(ascii_reads, k, rc))


@pytest.mark.parametrize("world,k,rc,n,L", [(3, 31, True, 1500, 150), (8, 31, True, 2500, 150), (2, 21, False, 900, 100), (4, 15, True, 700, 53),
                                            (3, 31, True, 400, 31), (8, 27, False, 1200, 101), (5, 31, True, 64, 150)])
def test_supermer_route_of_the_sharded_build(oracle, monkeypatch, world, k, rc, n, L):
    """The route of three ranks and more by packed key (dist.hip, supermer.hip): every read is cut into runs of windows whose cores
    share a minimizer, each run travels ONCE as a 16-byte record to the rank a hash of the minimizer names, and every rank counts what
    it received -- distinct supermers, then their k-mers.  Reads with N (skipped whole), reads of exactly k bases (one window), ranks
    without reads; against the oracle's sequential build: the edge multiset, every k-mer on one rank, every node owned once, end
    points and labels"""
    from katome_amd.build import GpuGraph
    monkeypatch.setenv("KATOME_DIST_ROUTE", "supermers")
    ascii_reads, packed, skip = _reads(oracle, n, L, 9000, 4e-3, 3)
    g, rb = GpuGraph.create_from_packed(packed, n, L, skip=skip, reverse_complement=rc, k=k, n_devices=world, ranks_share_device=True)
    ref = oracle.build_ascii(ascii_reads, k, rc)
    assert rb == ref.read_bytes
    assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges)
    assert g.multiset() == ref.multiset()
    ek, nk = g.key_ints("edge"), g.key_ints("node")
    assert len(set(ek)) == len(ek) and len(set(nk)) == len(nk)
    mask = (1 << (2 * (k - 1))) - 1
    for e in range(g.n_edges):
        assert nk[int(g.edge_src[e])] == ek[e] >> 2 and nk[int(g.edge_dst[e])] == ek[e] & mask
    assert set(g.edge_src.tolist()) | set(g.edge_dst.tolist()) == set(range(g.n_nodes))


def test_route_names(oracle, monkeypatch):
    """katome_dist_route: what a builder will send -- "local" up to two ranks, KATOME_DIST_ROUTE overrides (supermers only by packed key)"""
    from katome_amd import _lib
    from katome_amd import shard as ks
    comm = ks.Comm.rccl(0, 1, 0)
    try:
        for env, fs, want in ((None, False, b"local"), ("tiles", False, b"tiles"), ("supermers", False, b"supermers (if the reads allow)"),
                              ("supermers", True, b"tiles")):
            if env:
                monkeypatch.setenv("KATOME_DIST_ROUTE", env)
            b = ks.ShardedBuilder(comm, 31, True, 0, first_seen_order=fs)
            assert _lib.lib().katome_dist_route(b._h) == want, (env, fs)
            b.close()
    finally:
        comm.close()


def test_rccl_transport_at_world_size_one(oracle):
    """RCCL itself (ncclCommInitRank from a unique id, grouped send/recv to self, allreduce): the process-per-GPU route
    of bench.py with one rank -- same graph as the oracle's, exchange accounting readable"""
    from katome_amd import _lib
    L_ = _lib.lib()
    ident = (C.c_uint8 * 128)()
    assert L_.katome_comm_unique_id(ident) == 0, _lib.last_error()
    comm = C.c_void_p()
    assert L_.katome_comm_create_rccl(ident, 0, 1, 0, C.byref(comm)) == 0, _lib.last_error()
    assert L_.katome_comm_kind(comm) == b"rccl" and L_.katome_comm_world(comm) == 1
    k, rc, n, L = 31, True, 1000, 150
    ascii_reads, packed, skip = _reads(oracle, n, L, 20000, 5e-3, 2)
    pt, st = torch.from_numpy(np.concatenate([packed, np.zeros(32, np.uint8)])).cuda(), torch.from_numpy(skip).cuda()
    d, g = _dist_build_one_process(k, rc, pt, st, n, L, False, comm)
    ref = oracle.build_ascii(ascii_reads, k, rc)
    assert (g.total_nodes, g.total_edges) == (ref.n_nodes, ref.n_edges) == (g.n_nodes, g.n_edges)
    w = torch.empty(g.n_edges, dtype=torch.int32, device="cuda")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(w.data_ptr()), C.c_void_p(g.d_edge_weight), C.c_size_t(g.n_edges * 4), 3)
    assert int(w.to(torch.int64).sum().item()) == int(ref.edge_weight.astype(np.uint64).sum())
    nx = L_.katome_dist_exchange_count()
    stats = (C.c_uint64 * (4 * nx))()
    assert L_.katome_dist_exchange_read(d, stats) == 0
    names = [L_.katome_dist_exchange_name(i).decode() for i in range(nx)]
    assert "exchange_kmers" in names and stats[4 * names.index("exchange_kmers")] > 0      # calls were counted (one rank is "few ranks": its own reads are counted locally, distinct k-mers routed)
    L_.katome_dist_destroy(d)
    L_.katome_comm_destroy(comm)


# ---- one PROCESS per rank (bench.py's shape), several of them sharing this box's one GPU ------------------------------
def _process_rank(rank, world, port, k, rc, n_reads, read_len, first_seen, prune, out_dir):
    """HIP kernels for everything; the exchanges travel through torch.distributed/gloo (RCCL refuses two ranks on one GPU)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from katome_amd import shard as ks
    from oracle import oracle as o
    from helpers import pack_reads_ascii as pack
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        ascii_reads = o.synth_reads(0, n_reads, read_len, 60000, 2e-3, 2)
        has_n = (ascii_reads == ord("N")).any(axis=1)
        clean = ascii_reads.copy()
        clean[clean == ord("N")] = ord("A")
        first, count = ks.shard_range(n_reads, world, rank)
        packed = torch.from_numpy(np.concatenate([pack(clean[first:first + count]).reshape(-1), np.zeros(32, np.uint8)])).cuda()
        skip = torch.from_numpy(np.concatenate([has_n[first:first + count].astype(np.uint8), np.zeros(16, np.uint8)])).cuda()
        comm = ks.Comm.over_torch(device=0)
        assert comm.kind == "callbacks" and (comm.rank, comm.world) == (rank, world)
        b = ks.ShardedBuilder(comm, k, rc, 0, first_seen_order=first_seen)
        b.add_reads(packed, first, count, read_len, skip, batch_reads=2048)
        g = b.finalize()
        out = dict(total_nodes=g.total_nodes, total_edges=g.total_edges, node_base=g.node_base,
                   edge_key=g.edge_key.cpu().numpy().view(np.uint64), weight=g.edge_weight.cpu().numpy().view(np.uint32),
                   src=g.edge_src.cpu().numpy(), dst=g.edge_dst.cpu().numpy(), node_key=g.node_key.cpu().numpy().view(np.uint64))
        if first_seen:
            out.update(edge_id=g.edge_id.cpu().numpy(), node_id=g.node_id.cpu().numpy(), label=g.edge_label.cpu().numpy())
        if first_seen and prune == 2:              # remove_dead_paths on the sharded graph: every rank keeps its share
            del g
            pg, st = b.remove_dead_paths()
            out.update(p_total_nodes=pg.total_nodes, p_total_edges=pg.total_edges, p_edge_id=pg.edge_id.cpu().numpy(),
                       p_node_id=pg.node_id.cpu().numpy(), p_label=pg.edge_label.cpu().numpy(), p_src=pg.edge_src.cpu().numpy(),
                       p_dst=pg.edge_dst.cpu().numpy(), p_weight=pg.edge_weight.cpu().numpy().view(np.uint32),
                       p_age=pg.edge_age.cpu().numpy(), p_passes=st["passes"], p_removed=st["removed_edges"])
            del pg
        elif first_seen:
            root = b.gather(0)
            assert (root is not None) == (rank == 0)
            if root is not None:
                dg = root.remove_dead_paths()[0] if prune else root.graph()
                out.update(root_src=dg.edge_src.cpu().numpy(), root_dst=dg.edge_dst.cpu().numpy(),
                           root_weight=dg.edge_weight.cpu().numpy().view(np.uint32), root_label=dg.edge_label.cpu().numpy(),
                           root_nodes=dg.n_nodes)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
        b.close()
        comm.close()
    finally:
        dist.destroy_process_group()


def _spawn(world, *args):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_process_rank, args=(world, port) + args, nprocs=world, join=True)


@pytest.mark.parametrize("world,k,rc,L", [(2, 31, True, 150), (3, 31, False, 100), (2, 63, True, 150), (4, 40, True, 103)])
def test_process_per_rank_by_packed_key(oracle, tmp_path, world, k, rc, L):      # (the test runner holds the card too: six processes at most)
    n_reads = 9000
    _spawn(world, k, rc, n_reads, L, False, False, str(tmp_path))
    ref = oracle.build_ascii(oracle.synth_reads(0, n_reads, L, 60000, 2e-3, 2), k, rc)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    nw = 1 if 2 * k <= 62 else 2

    def to_int(row):
        return int(row[0]) if len(row) == 1 else (int(row[0]) << 64) | int(row[1])
    node_of, merged = {}, {}
    mask = (1 << (2 * (k - 1))) - 1
    for p in parts:
        for i, row in enumerate(p["node_key"].reshape(-1, nw)):
            node_of[int(p["node_base"]) + i] = to_int(row)
    for p in parts:
        assert (int(p["total_nodes"]), int(p["total_edges"])) == (ref.n_nodes, ref.n_edges)
        keys = [to_int(r) for r in p["edge_key"].reshape(-1, nw)]
        assert keys == sorted(keys)
        for j, (key, w) in enumerate(zip(keys, p["weight"])):
            assert key not in merged
            merged[key] = int(w)
            if j % 53 == 0:
                assert node_of[int(p["src"][j])] == key >> 2 and node_of[int(p["dst"][j])] == key & mask
    from helpers import kmer_to_int
    assert sorted(merged.items()) == sorted((kmer_to_int(s), w) for s, w in ref.multiset())
    assert sorted(node_of) == list(range(ref.n_nodes)) and len(set(node_of.values())) == ref.n_nodes


@pytest.mark.parametrize("world,k,rc,L,prune", [(2, 31, True, 150, False), (3, 63, True, 150, True), (2, 40, False, 103, True)])
def test_process_per_rank_in_reference_order(oracle, tmp_path, world, k, rc, L, prune):
    """the ranks' shares carry the reference's indices (every index exactly once across ranks, the right edge at each);
    gathered to rank 0 -- and pruned there -- the arrays equal the oracle's petgraph index for index"""
    n_reads = 6000
    _spawn(world, k, rc, n_reads, L, True, prune, str(tmp_path))
    ascii_reads = oracle.synth_reads(0, n_reads, L, 60000, 2e-3, 2)
    ref = oracle.build_ascii(ascii_reads, k, rc)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    label = np.zeros_like(ref.edge_label)
    src, dst = np.full(ref.n_edges, -1, np.int64), np.full(ref.n_edges, -1, np.int64)
    seen_e, seen_n = np.zeros(ref.n_edges, np.int32), np.zeros(ref.n_nodes, np.int32)
    for p in parts:
        assert (int(p["total_nodes"]), int(p["total_edges"])) == (ref.n_nodes, ref.n_edges)
        ids = p["edge_id"]
        np.add.at(seen_e, ids, 1)
        np.add.at(seen_n, p["node_id"], 1)
        label[ids], src[ids], dst[ids] = p["label"], p["src"], p["dst"]
    assert (seen_e == 1).all() and (seen_n == 1).all()
    assert np.array_equal(label, ref.edge_label)
    assert np.array_equal(src.astype(np.uint64), ref.edge_src) and np.array_equal(dst.astype(np.uint64), ref.edge_dst)
    want = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=True) if prune else ref
    r0 = parts[0]
    assert int(r0["root_nodes"]) == want.n_nodes
    assert np.array_equal(r0["root_label"], want.edge_label) and np.array_equal(r0["root_weight"], want.edge_weight)
    assert np.array_equal(r0["root_src"].astype(np.uint64), want.edge_src) and np.array_equal(r0["root_dst"].astype(np.uint64), want.edge_dst)


@pytest.mark.parametrize("world,k,rc,L", [(2, 31, True, 150), (3, 63, True, 150), (4, 40, False, 103)])
def test_process_per_rank_pruned_without_a_gather(oracle, tmp_path, world, k, rc, L):
    """one process per rank (bench.py's shape; exchanges over gloo), katome_dist_remove_dead_paths on the sharded graph: the
    ranks' shares put side by side by their indices are the oracle's pruned petgraph -- every index exactly once, the right
    edge, end points, weight and age at each"""
    n_reads = 6000
    _spawn(world, k, rc, n_reads, L, True, 2, str(tmp_path))
    ascii_reads = oracle.synth_reads(0, n_reads, L, 60000, 2e-3, 2)
    want = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=True)
    full = oracle.build_ascii(ascii_reads, k, rc)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert want.n_edges < full.n_edges and (want.n_edges > 0 or k == 63)      # (k = 63 at this error rate: every path is shorter than 2k)
    label = np.zeros_like(want.edge_label)
    src, dst = np.full(want.n_edges, -1, np.int64), np.full(want.n_edges, -1, np.int64)
    weight, age = np.zeros(want.n_edges, np.uint32), np.zeros(want.n_edges, np.int64)
    seen_e, seen_n = np.zeros(want.n_edges, np.int32), np.zeros(want.n_nodes, np.int32)
    for p in parts:
        assert (int(p["p_total_nodes"]), int(p["p_total_edges"])) == (want.n_nodes, want.n_edges)
        assert int(p["p_removed"]) == full.n_edges - want.n_edges and int(p["p_passes"]) >= 2
        ids = p["p_edge_id"]
        np.add.at(seen_e, ids, 1)
        np.add.at(seen_n, p["p_node_id"], 1)
        label[ids], src[ids], dst[ids], weight[ids], age[ids] = p["p_label"], p["p_src"], p["p_dst"], p["p_weight"], p["p_age"]
    assert (seen_e == 1).all() and (seen_n == 1).all()
    assert np.array_equal(label, want.edge_label) and np.array_equal(weight, want.edge_weight)
    assert np.array_equal(src.astype(np.uint64), want.edge_src) and np.array_equal(dst.astype(np.uint64), want.edge_dst)
    assert np.array_equal(age.astype(np.uint64) + 1, want.edge_slot)


_SORTED_SHARDED_SCRIPT = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from helpers import pack_reads_ascii, int_to_kmer
from oracle import oracle as o
from katome_amd.build import GpuGraph
from katome_amd import shard as ks
for world, k, rc, n, L in ((2, 31, True, 1200, 150), (3, 21, False, 900, 100), (8, 31, True, 1500, 150), (4, 15, True, 700, 53)):
    reads = o.synth_reads(11, n, L, 9000, 4e-3, 0)
    packed = pack_reads_ascii(reads)
    g, _ = GpuGraph.create_from_packed(packed, n, L, skip=None, reverse_complement=rc, k=k, n_devices=world, ranks_share_device=True)
    ref = o.build_ascii(reads, k, rc)
    assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges), (world, k, g.n_nodes, g.n_edges, ref.n_nodes, ref.n_edges)
    assert g.multiset() == ref.multiset(), (world, k)
    print("SS", world, k, g.n_nodes, g.n_edges, hashlib.sha256(repr(g.multiset()).encode()).hexdigest())
# Clean::remove_weak_edges as the edges are counted (one rank over RCCL: the route of bench.py)
k, rc, n, L, thr = 31, True, 2000, 150, 2
reads = o.synth_reads(12, n, L, 4000, 4e-3, 0)
pt = torch.from_numpy(np.concatenate([pack_reads_ascii(reads).reshape(-1), np.zeros(32, np.uint8)])).cuda()
comm = ks.Comm.rccl(0, 1, 0)
comm.set_max_message_bytes(1 << 16)            # many slices: the collected records arrive piecemeal
sb = ks.ShardedBuilder(comm, k, rc)
sb.remove_weak_edges(thr)
sb.add_reads(pt, 0, n, L, None, 0)
g = sb.finalize()
ref = o.build_ascii(reads, k, rc, remove_weak_edges=thr)
got = sorted(zip((int_to_kmer(int(v) & ((1 << 64) - 1), k) for v in g.edge_key.cpu().numpy().reshape(-1).tolist()),
                 (int(w) for w in g.edge_weight.cpu().numpy().view(np.uint32))))
assert got == ref.multiset() and len(got) > 100, (len(got), ref.n_edges)
print("SS weak", len(got), hashlib.sha256(repr(got).encode()).hexdigest())
del g
sb.close(); comm.close()
"""


@pytest.mark.parametrize("route", ["local", "tiles", "supermers"])
def test_sharded_routes_with_the_last_level_counted_by_sorting(tmp_path, route):
    """by packed key the k-mer records that reach their owners are kept and counted by sorting (from 4 M records on; forced here
    with KATOME_SORTED_COUNT=2, in a process of its own: the switch is read once) -- same graph as the oracle's and as the
    k-mer table's (KATOME_SORTED_COUNT=0), with and without the remove_weak_edges threshold, ranks without reads included"""
    import subprocess
    script = tmp_path / "ss.py"
    script.write_text(_SORTED_SHARDED_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra):
        env = dict(os.environ, KATOME_DIST_ROUTE=route, **extra)
        out = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        return sorted(line for line in out.stdout.splitlines() if line.startswith("SS "))
    forced = run({"KATOME_SORTED_COUNT": "2"})
    assert len(forced) == 5
    assert run({"KATOME_SORTED_COUNT": "0"}) == forced


def test_sharded_build_at_a_size_where_the_routes_run_in_earnest():
    """8 thread ranks on 8 M reads through the supermer route and the level-by-level route (the default there) against the one-GPU build of the same reads (order-free
    checksums of edges, weights, nodes; every edge's end points) -- and in reasonable time: while a tile's owner was the hash
    range its table slot came from, every rank crowded its keys into an eighth of its tables and this build took 40 s"""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_sharded_scale.py"), "8000000", "8", "supermers,tiles"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "same: True" in out.stdout and "OK" in out.stdout.splitlines()[-1]
    took = [float(line.split()[-2]) for line in out.stdout.splitlines() if line.startswith("8 ranks")]
    assert len(took) == 2 and max(took) < 15.0, out.stdout
    assert time.time() - t0 < 150
