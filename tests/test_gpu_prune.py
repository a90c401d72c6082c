"""Prunable::remove_dead_paths (pruner.rs:36-82) through the C ABI, against the oracle's petgraph restatement:
the pruned graph must come back index for index (edge order, node ids), which covers the reference's quirks --
first_edge = most recently added live edge, walks stopping only at in-degree >= 3, swap_remove re-numbering, and an
index collected by two walks removing whatever edge was swapped in."""
import json
import os

import numpy as np
import pytest

from helpers import pack_reads_ascii

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _same(g, ref, k):
    assert (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges)
    if getattr(g, "edge_age", None) is not None:
        # the oracle hands SEQUENCES slots out in the order edges are first added (pt_graph.rs:176-194): slot = age + 1
        assert np.array_equal(g.edge_age.astype(np.uint64) + 1, ref.edge_slot)
    assert np.array_equal(g.edge_label, ref.edge_label)
    assert np.array_equal(g.edge_weight, ref.edge_weight)
    assert np.array_equal(g.edge_src, ref.edge_src) and np.array_equal(g.edge_dst, ref.edge_dst)
    # node keys follow their nodes
    ek, nk = g.key_ints("edge"), g.key_ints("node")
    mask = (1 << (2 * (k - 1))) - 1
    for e in range(g.n_edges):
        assert nk[int(g.edge_src[e])] == ek[e] >> 2 and nk[int(g.edge_dst[e])] == ek[e] & mask


@pytest.mark.parametrize("i", [0, 1, 2])
def test_reference_pinned_counts(golden_dir, i):
    """tests/pruner.rs:205-216 (removes_dead_paths) with its expected counts (tests/pruner.rs:37-150: 0 / 0 each)"""
    from katome_amd.build import GpuGraph, InputFileType, set_global_k_sizes
    pinned = json.load(open(os.path.join(golden_dir, "pinned.json")))
    set_global_k_sizes(pinned["k"])
    g, _ = GpuGraph.create([os.path.join(golden_dir, pinned["fixtures"][i])], InputFileType.Fastq, False, 0,
                           first_seen_order=True, remove_dead_paths=True)
    want = pinned["remove_dead_paths"]["counts"][i]
    assert [g.n_nodes, g.n_edges] == want
    assert [g.stats().node_count, g.stats().edge_count] == want


@pytest.mark.parametrize("name,k,rc", [("data2.txt", 40, True), ("data3.txt", 31, True), ("data2.txt", 16, True),
                                       ("data3.txt", 12, False), ("data3.txt", 8, True), ("data2.txt", 6, True),
                                       ("data3.txt", 5, False)])
def test_fixtures_against_oracle(oracle, golden_dir, name, k, rc):
    from katome_amd.build import GpuGraph, InputFileType, set_global_k_sizes
    path = os.path.join(golden_dir, name)
    set_global_k_sizes(k)
    g, _ = GpuGraph.create([path], InputFileType.Fastq, rc, 0, first_seen_order=True, remove_dead_paths=True)
    _same(g, oracle.build_files([path], k, rc, remove_dead_paths=True), k)


def _random_reads(seed, n_reads, read_len, genome_len, err):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len)
    reads = np.zeros((n_reads, read_len), np.uint8)
    for i in range(n_reads):
        s = rng.integers(0, genome_len - read_len + 1)
        r = genome[s:s + read_len].copy()
        m = rng.random(read_len) < err
        r[m] = rng.integers(0, 4, int(m.sum()))
        reads[i] = np.frombuffer(b"ACGT", np.uint8)[r]
    return reads


@pytest.mark.parametrize("host_replays", [False, True, "no slots", "walk all", "check walks"])
@pytest.mark.parametrize("seed", range(24))
def test_random_graphs_against_oracle(oracle, monkeypatch, seed, host_replays):
    """small k and noisy reads: branching, cycles, self-loops, merging tips (duplicate indices), several passes.
    host_replays: the sequential statement of the two index replays (prune_replay.h) instead of their device forms --
    the route a pass takes when its node moves chain further than the device form follows"""
    from katome_amd import device as kd
    if host_replays == "no slots":       # without the per-vertex edge slots every pass streams all edges (the low-memory route)
        monkeypatch.setenv("KATOME_PRUNE_NO_SLOTS", "1")
        host_replays = False
    elif host_replays == "walk all":     # every listed Input vertex is walked in every pass, as the reference does
        monkeypatch.setenv("KATOME_PRUNE_WALK_ALL", "1")
        host_replays = False
    elif host_replays == "check walks":  # ... and the call fails if a walk the change tracking would skip is dead
        monkeypatch.setenv("KATOME_PRUNE_CHECK_WALKS", "1")
        host_replays = False
    elif host_replays:
        monkeypatch.setenv("KATOME_PRUNE_HOST_EDGES", "1")
        monkeypatch.setenv("KATOME_PRUNE_HOST_NODES", "1")
    k = [4, 5, 6, 8, 11, 17][seed % 6]
    L = k + 3 + seed % 7
    reads = _random_reads(seed, 30 + 40 * seed, L, 60 + 50 * seed, 0.04)
    packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
    for rc in (False, True):
        b = kd.Builder(k, rc, first_seen_order=True)
        span = b.tile_span(L)
        if span > 1:
            b.insert_tiles(b.extract_tiles(packed, len(reads), L, span), span)
        else:
            b.insert(b.extract_fixed(packed, len(reads), L))
        before = b.finalize()
        ref0 = oracle.build_ascii(reads, k, rc)
        assert (before.n_nodes, before.n_edges) == (ref0.n_nodes, ref0.n_edges)
        dg, st = b.remove_dead_paths()
        ref = oracle.build_ascii(reads, k, rc, remove_dead_paths=True)
        assert (dg.n_nodes, dg.n_edges) == (ref.n_nodes, ref.n_edges)
        assert np.array_equal(dg.edge_label.cpu().numpy(), ref.edge_label)
        assert np.array_equal(dg.edge_weight.cpu().numpy().view(np.uint32), ref.edge_weight)
        assert np.array_equal(dg.edge_src.cpu().numpy().view(np.uint64), ref.edge_src)
        assert np.array_equal(dg.edge_dst.cpu().numpy().view(np.uint64), ref.edge_dst)
        assert st["removed_edges"] == ref0.n_edges - ref.n_edges and st["removed_nodes"] == ref0.n_nodes - ref.n_nodes
        assert st["passes"] >= 1 and st["marked"] >= st["removed_edges"] - st["removed_by_duplicates"]
        assert (st["host_ms"] > 0) == (host_replays and st["removed_edges"] > 0)
        b.close()


def test_quirks_are_exercised(oracle):
    """the random family must actually reach the duplicate-index rule and need more than two passes somewhere"""
    from katome_amd import device as kd
    dup = passes = 0
    for seed in (1, 2, 7, 8, 13, 14):
        k = [4, 5, 6, 8, 11, 17][seed % 6]
        L = k + 3 + seed % 7
        reads = _random_reads(seed, 30 + 40 * seed, L, 60 + 50 * seed, 0.04)
        packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
        b = kd.Builder(k, True, first_seen_order=True)
        span = b.tile_span(L)
        if span > 1:
            b.insert_tiles(b.extract_tiles(packed, len(reads), L, span), span)
        else:
            b.insert(b.extract_fixed(packed, len(reads), L))
        b.finalize()
        _, st = b.remove_dead_paths()
        dup += st["removed_by_duplicates"]
        passes = max(passes, st["passes"])
        b.close()
    assert dup > 0 and passes > 2


@pytest.mark.parametrize("k,rc,n,L,glen", [(31, True, 30000, 100, 100000), (21, False, 40000, 80, 100000),
                                           (40, True, 20000, 103, 50000)])
def test_synthetic_workload_against_oracle(oracle, k, rc, n, L, glen):
    """sequencing-like input (coverage ~30x, 0.5% errors, some reads with N), ~15 passes.  (The reference's rule also
    cuts the genome wherever a walk meets a vertex with three incoming edges, so little survives at rc=True.)"""
    from katome_amd.build import GpuGraph
    ascii_reads = oracle.synth_reads(0, n, L, glen, 5e-3, 1)
    has_n = (ascii_reads == ord("N")).any(axis=1)
    clean = ascii_reads.copy()
    clean[clean == ord("N")] = ord("A")
    g, _ = GpuGraph.create_from_packed(pack_reads_ascii(clean).reshape(-1), n, L, skip=has_n.astype(np.uint8),
                                       reverse_complement=rc, k=k, first_seen_order=True, remove_dead_paths=True)
    ref = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=True)
    full = oracle.build_ascii(ascii_reads, k, rc)
    assert 0 < ref.n_edges < full.n_edges                  # the case prunes something and keeps something
    _same(g, ref, k)


def test_needs_first_seen_order(golden_dir):
    from katome_amd.build import GpuGraph, InputFileType, KatomePanic, set_global_k_sizes
    set_global_k_sizes(40)
    with pytest.raises(KatomePanic) as e:
        GpuGraph.create([os.path.join(golden_dir, "data1.txt")], InputFileType.Fastq, False, 0, remove_dead_paths=True)
    assert e.value.name == "E_ARG"


def test_before_finalize_is_an_error():
    from katome_amd import device as kd
    from katome_amd.build import KatomePanic
    b = kd.Builder(21, True, first_seen_order=True)
    with pytest.raises(KatomePanic):
        b.remove_dead_paths()
    b.close()


def _variable_length_fastq(path, seed, n_reads, k, genome_len=3000, err=0.01):
    """reads of every length from k to ~5k off one genome (so they overlap), a few with an N, both header styles"""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len)
    lines = []
    for i in range(n_reads):
        n = int(rng.integers(k, 5 * k + 7))
        s0 = int(rng.integers(0, genome_len - n + 1))
        r = genome[s0:s0 + n].copy()
        m = rng.random(n) < err
        r[m] = rng.integers(0, 4, int(m.sum()))
        s = "".join("ACGT"[c] for c in r)
        if i % 23 == 5:
            s = s[:3] + "N" + s[4:]
        lines += ["@r%d" % i, s, "+", "I" * n]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


@pytest.mark.parametrize("tiles", [True, False])
@pytest.mark.parametrize("k,rc,n_reads", [(11, True, 900), (12, False, 700), (31, True, 1500), (40, True, 800), (5, True, 200),
                                          (63, True, 400), (47, False, 500)])        # tiles of 64..95 bases (three words)
def test_variable_length_reads_in_reference_order(oracle, tmp_path, monkeypatch, k, rc, n_reads, tiles):
    """first-seen numbering and remove_dead_paths for reads of unequal length (the general file route: one record
    per window, sequence numbers from the per-read window prefix), in one batch and in many"""
    from katome_amd.build import GpuGraph, InputFileType, set_global_k_sizes
    fq = str(tmp_path / "var.fq")
    _variable_length_fastq(fq, 3 * k + n_reads, n_reads, k)
    set_global_k_sizes(k)
    if n_reads % 200:
        monkeypatch.setenv("KATOME_VAR_BATCH_RECORDS", "1500")
    if not tiles:                          # every window on its own instead of tiles + left-over windows
        monkeypatch.setenv("KATOME_NO_TILES", "1")
    g, rb = GpuGraph.create([fq], InputFileType.Fastq, rc, 0, first_seen_order=True)
    ref = oracle.build_files([fq], k, rc)
    assert rb == ref.read_bytes
    _same(g, ref, k)
    g, _ = GpuGraph.create([fq], InputFileType.Fastq, rc, 0, first_seen_order=True, remove_dead_paths=True)
    _same(g, oracle.build_files([fq], k, rc, remove_dead_paths=True), k)


def _dev_arrays(dg):
    return (dg.edge_label.cpu().numpy(), dg.edge_weight.cpu().numpy().view(np.uint32),
            dg.edge_src.cpu().numpy().view(np.uint64), dg.edge_dst.cpu().numpy().view(np.uint64))


def _assert_dev_same(dg, ref):
    assert (dg.n_nodes, dg.n_edges) == (ref.n_nodes, ref.n_edges)
    if dg.edge_age is not None:
        assert np.array_equal(dg.edge_age.cpu().numpy().view(np.uint32).astype(np.uint64) + 1, ref.edge_slot)
    lab, w, s, d = _dev_arrays(dg)
    assert np.array_equal(lab, ref.edge_label) and np.array_equal(w, ref.edge_weight)
    assert np.array_equal(s, ref.edge_src) and np.array_equal(d, ref.edge_dst)


@pytest.mark.parametrize("k,rc,thr", [(31, True, 2), (12, False, 3), (40, True, 2), (6, True, 30), (21, True, 4)])
def test_remove_weak_edges_in_reference_order(oracle, k, rc, thr):
    """Clean::remove_weak_edges on a first-seen-order builder: petgraph's retain_edges / retain_nodes numbering
    (descending swap_removes), asked for before finalize, after it, and around remove_dead_paths in both orders"""
    from katome_amd import device as kd
    n, L = 2500, 110
    ascii_reads = oracle.synth_reads(0, n, L, 15000, 1e-2, 0)
    packed = torch.from_numpy(pack_reads_ascii(ascii_reads).reshape(-1).copy()).cuda()

    def build():
        b = kd.Builder(k, rc, first_seen_order=True)
        span = b.tile_span(L)
        if span > 1:
            b.insert_tiles(b.extract_tiles(packed, n, L, span), span)
        else:
            b.insert(b.extract_fixed(packed, n, L))
        return b

    full = oracle.build_ascii(ascii_reads, k, rc)
    want_w = oracle.build_ascii(ascii_reads, k, rc, remove_weak_edges=thr)
    assert 0 < want_w.n_edges < full.n_edges
    b = build()
    b.remove_weak_edges(thr)                       # before finalize
    _assert_dev_same(b.finalize(), want_w)
    dg, _ = b.remove_dead_paths()                  # ... then dead paths
    _assert_dev_same(dg, oracle.build_ascii(ascii_reads, k, rc, remove_weak_edges=thr, remove_dead_paths=True, stages="wd"))
    b.close()
    b = build()
    b.finalize()
    b.remove_weak_edges(thr)                       # after finalize
    _assert_dev_same(b.graph(), want_w)
    b.close()
    b = build()
    b.finalize()
    b.remove_dead_paths()
    b.remove_weak_edges(thr)                       # the assembler's order (asm/basic_assembler.rs:58-66, without the shrink between)
    _assert_dev_same(b.graph(), oracle.build_ascii(ascii_reads, k, rc, remove_weak_edges=thr, remove_dead_paths=True, stages="dw"))
    b.close()


@pytest.mark.parametrize("k,rc,thr,glen", [(31, True, 2, 15000), (21, False, 2, 15000), (12, True, 3, 4000), (40, True, 2, 20000),
                                           (8, True, 4, 1500)])
def test_every_stage_up_to_collapse(oracle, k, rc, thr, glen):
    """assemble_with_graph (asm/basic_assembler.rs:58-72) on the device, stage by stage against the oracle: remove_dead_paths,
    standardize_contigs, remove_weak_edges(threshold), standardize_contigs, standardize_edges(genome length, k, threshold),
    remove_dead_paths -- the graph `collapse` receives, index for index (weights, labels, end points, ages)"""
    from katome_amd import device as kd
    n, L = 2500, 110
    ascii_reads = oracle.synth_reads(0, n, L, glen, 8e-3, 0)
    packed = torch.from_numpy(pack_reads_ascii(ascii_reads).reshape(-1).copy()).cuda()
    b = kd.Builder(k, rc, first_seen_order=True)
    b.count_reads(packed, n, L)
    b.finalize()
    oracle.set_genome_length(glen)

    def ref(stages):
        return oracle.build_ascii(ascii_reads, k, rc, remove_weak_edges=thr, stages=stages)

    b.remove_dead_paths()
    b.standardize_contigs()
    _assert_dev_same(b.graph(), ref("dc"))
    b.remove_weak_edges(thr)
    b.standardize_contigs()
    _assert_dev_same(b.graph(), ref("dcwc"))
    b.standardize_edges(glen, thr)
    _assert_dev_same(b.graph(), ref("dcwce"))
    dg, _ = b.remove_dead_paths()
    final = ref("dcwced")
    _assert_dev_same(dg, final)
    full = oracle.build_ascii(ascii_reads, k, rc)
    assert final.n_edges < full.n_edges
    b.close()


@pytest.mark.parametrize("k,rc", [(21, True), (9, True), (16, False)])
def test_standardize_contigs_on_any_numbering(oracle, golden_dir, k, rc):
    """no re-numbering is involved, so the default (by packed key) order gets the same weight per k-mer"""
    from katome_amd import device as kd
    from katome_amd.build import ingest_files, InputFileType
    path = os.path.join(golden_dir, "data3.txt")
    r = ingest_files([path], InputFileType.Fastq, k)
    packed = torch.from_numpy(r["packed"].copy()).cuda()
    b = kd.Builder(k, rc)
    b.count_reads(packed, r["n_reads"], r["fixed_len"])
    b.finalize()
    b.standardize_contigs()
    dg = b.graph()
    nw = dg.key_words
    ek = dg.edge_key.cpu().numpy().view(np.uint64).reshape(-1, nw)
    keys = [int(x[0]) if nw == 1 else (int(x[0]) << 64) | int(x[1]) for x in ek]
    from helpers import int_to_kmer
    got = sorted(zip([int_to_kmer(v, k) for v in keys], dg.edge_weight.cpu().numpy().view(np.uint32).tolist()))
    assert got == oracle.build_files([path], k, rc, stages="c").multiset()
    b.close()


def _collapse_from_device_graph(oracle, dg, k):
    """rebuild a PtGraph from the device arrays the way INTEGRATION.md prescribes (edges added in ascending age, so that
    petgraph's adjacency lists come out as the reference has them) and run the oracle's restated collapse on it"""
    from helpers import int_to_kmer
    nw = dg.key_words
    ek = dg.edge_key.cpu().numpy().view(np.uint64).reshape(-1, nw)
    seqs = [int_to_kmer(int(x[0]) if nw == 1 else (int(x[0]) << 64) | int(x[1]), k) for x in ek]
    src = dg.edge_src.cpu().numpy().view(np.uint64).tolist()
    dst = dg.edge_dst.cpu().numpy().view(np.uint64).tolist()
    w = dg.edge_weight.cpu().numpy().view(np.uint32).tolist()
    age = dg.edge_age.cpu().numpy().view(np.uint32).tolist() if dg.edge_age is not None else list(range(dg.n_edges))
    order = sorted(range(dg.n_edges), key=lambda e: age[e])
    edges = [(src[e], dst[e], w[e], i + 1) for i, e in enumerate(order)]
    slots = [None] + [seqs[e] for e in order]
    return oracle.run_from_edges(dg.n_nodes, edges, "C", 0, k, slots).collapsed


@pytest.mark.parametrize("i", [0, 1, 2])
def test_contigs_from_the_gpu_graph_pinned(oracle, golden_dir, i):
    """tests/collapser.rs:32 ([2, 92, 233] contigs) with the GRAPH coming from the GPU build (reference numbering) and only
    collapse run by the oracle: the same contigs, in the same order, as the all-CPU pipeline"""
    from katome_amd import device as kd
    from katome_amd.build import ingest_files, InputFileType
    pinned = json.load(open(os.path.join(golden_dir, "pinned.json")))
    path, k = os.path.join(golden_dir, pinned["fixtures"][i]), pinned["k"]
    r = ingest_files([path], InputFileType.Fastq, k)
    packed = torch.from_numpy(r["packed"].copy()).cuda()
    b = kd.Builder(k, False, first_seen_order=True)
    b.count_reads(packed, r["n_reads"], r["fixed_len"])
    dg = b.finalize()
    got = _collapse_from_device_graph(oracle, dg, k)
    assert len(got) == pinned["collapse"]["contigs"][i]
    assert got == oracle.build_files([path], k, False, stages="C").collapsed
    b.close()


@pytest.mark.parametrize("k,rc,thr,glen", [(31, True, 2, 15000), (21, False, 2, 15000), (12, True, 3, 4000), (8, True, 4, 1500)])
def test_contigs_of_the_whole_pipeline(oracle, k, rc, thr, glen):
    """assemble_with_graph end to end: every stage before collapse on the device, collapse itself by the oracle on the
    graph rebuilt from the device arrays -- the contigs equal those of the oracle's all-CPU pipeline, string for string and
    in the same order"""
    from katome_amd import device as kd
    n, L = 2500, 110
    ascii_reads = oracle.synth_reads(0, n, L, glen, 8e-3, 0)
    packed = torch.from_numpy(pack_reads_ascii(ascii_reads).reshape(-1).copy()).cuda()
    b = kd.Builder(k, rc, first_seen_order=True)
    b.count_reads(packed, n, L)
    b.finalize()
    b.remove_dead_paths()
    b.standardize_contigs()
    b.remove_weak_edges(thr)
    b.standardize_contigs()
    b.standardize_edges(glen, thr)
    dg, _ = b.remove_dead_paths()
    got = _collapse_from_device_graph(oracle, dg, k)
    oracle.set_genome_length(glen)
    want = oracle.build_ascii(ascii_reads, k, rc, remove_weak_edges=thr, stages="dcwcedC").collapsed
    assert len(want) > 10 and sum(map(len, want)) > 2000
    assert got == want
    b.close()


@pytest.mark.parametrize("name,k,rc,thr,glen,stages", [("data3.txt", 21, True, 2, 3000, "dcwced"), ("data2.txt", 12, False, 2, 2500, "dcwced"),
                                                       ("data3.txt", 40, False, 1, 20000, "cwe"), ("data2.txt", 9, True, 3, 800, "wdc")])
def test_host_entry_with_stages(oracle, golden_dir, name, k, rc, thr, glen, stages):
    """katome_build_files_staged: files in, the graph after the named stages out (host arrays, ages included)"""
    from katome_amd.build import GpuGraph, InputFileType, KatomePanic, set_global_k_sizes
    path = os.path.join(golden_dir, name)
    set_global_k_sizes(k)
    g, rb = GpuGraph.create([path], InputFileType.Fastq, rc, thr, first_seen_order=True, stages=stages, original_genome_length=glen)
    oracle.set_genome_length(glen)
    ref = oracle.build_files([path], k, rc, remove_weak_edges=thr, stages=stages)
    assert rb == ref.read_bytes
    _same(g, ref, k)
    with pytest.raises(KatomePanic):          # the stages walk petgraph's numbering
        GpuGraph.create([path], InputFileType.Fastq, rc, thr, stages=stages, original_genome_length=glen)


def _naive_edge_removals(pos, mult, n_edges):
    """remove_paths (pruner.rs:199-217): indices sorted descending, Graph::remove_edge = Vec::swap_remove"""
    arr = list(range(n_edges))
    victims, dups = [], 0
    for d, c in sorted(zip(pos, mult), reverse=True):
        for r in range(c):
            if d < len(arr):
                victims.append(arr[d])
                dups += r != 0
                arr[d] = arr[-1]
                arr.pop()
    return victims, {p: v for p, v in enumerate(arr) if v != p}, len(arr), dups


@pytest.mark.parametrize("seed", range(60))
def test_edge_removal_replay_on_the_device(seed):
    """katome_dev_replay_edge_removals (scan of n -> max(n - c, d) + pointer jumping) against a literal swap_remove loop:
    repeated indices, runs that eat the whole tail, entries that end by removing the last edge itself, chains of
    marked positions feeding marked positions; sizes across several workgroups of the scan"""
    from katome_amd import device as kd
    rng = np.random.default_rng(1000 + seed)
    n_edges = int(rng.integers(1, 60)) if seed % 4 == 0 else int(rng.integers(1000, 30000))
    style = seed % 5
    if style == 0:      # a dense block of marks at the very top: pointer chains, pops
        u = int(rng.integers(1, n_edges + 1))
        pos = np.arange(n_edges - u, n_edges)
        pos = pos[rng.random(u) < 0.8] if u > 1 else pos
    elif style == 1:    # every position marked
        pos = np.arange(n_edges)
    else:
        u = int(rng.integers(0, n_edges + 1)) if style == 2 else int(rng.integers(0, max(2, n_edges // 10)))
        pos = np.sort(rng.choice(n_edges, u, replace=False))
    hi = (2, 6, 40, 3, 2)[style]
    mult = rng.integers(1, hi + 1, len(pos))
    if len(pos) and seed % 7 == 0:
        mult[rng.integers(0, len(pos))] = n_edges + 5          # one index listed more often than there are edges
    want_v, want_moves, want_left, want_dups = _naive_edge_removals(pos.tolist(), mult.tolist(), n_edges)
    d_pos = torch.from_numpy(pos.astype(np.int64)).to(torch.int32).cuda()
    d_mult = torch.from_numpy(mult.astype(np.int64)).to(torch.int32).cuda()
    v, to, frm, left, dups = kd.replay_edge_removals(d_pos, d_mult, n_edges)
    assert left == want_left and dups == want_dups
    assert v.cpu().tolist() == want_v
    assert dict(zip(to.cpu().tolist(), frm.cpu().tolist())) == want_moves and to.numel() == len(want_moves)


def _naive_node_removals(die, n_nodes):
    """remove_single_node per endpoint, larger current index first (pruner.rs:206-215); remove_node = swap_remove"""
    arr = list(range(n_nodes))
    where = {v: v for v in arr}
    for a, b in die:
        gone = sorted((v for v in (a, b) if v >= 0), key=lambda v: where[v], reverse=True)
        for v in gone:
            p = where.pop(v)
            last = arr.pop()
            if p < len(arr):
                arr[p] = last
                where[last] = p
    return {p: v for p, v in enumerate(arr) if v != p}, len(arr)


@pytest.mark.parametrize("seed", range(60))
def test_node_removal_replay_on_the_device(seed):
    """katome_dev_replay_node_removals (holes of the tail positions settled in rounds) against a literal swap_remove loop:
    pairs and single endpoints, most of the graph dying, removals biased to the tail (nodes that move several times)"""
    from katome_amd import device as kd
    rng = np.random.default_rng(5000 + seed)
    n_nodes = int(rng.integers(2, 80)) if seed % 4 == 0 else int(rng.integers(1000, 40000))
    style = seed % 5
    if style == 0:
        n_die = n_nodes                                                    # everything goes
    elif style == 1:
        n_die = int(rng.integers(n_nodes // 2, n_nodes + 1))
    else:
        n_die = int(rng.integers(0, n_nodes // 3 + 1))
    if style == 3:      # the dying nodes sit at the top and leave from the bottom of that block up: long chains of moves
        dying = list(range(n_nodes - n_die, n_nodes))[::-1]
    elif style == 4:    # descending from the top: every removal takes the last node itself
        dying = list(range(n_nodes - n_die, n_nodes))
    else:
        dying = rng.permutation(n_nodes)[:n_die].tolist()
    die = []
    while dying:
        kind = rng.integers(0, 4)
        if kind == 0 and len(dying) >= 2:
            die.append((dying.pop(), dying.pop()))
        elif kind == 1:
            die.append((dying.pop(), -1))
        elif kind == 2:
            die.append((-1, dying.pop()))
        else:
            die.append((-1, -1))
    want_moves, want_left = _naive_node_removals(die, n_nodes)
    d_die = torch.tensor(die, dtype=torch.int32).reshape(-1).cuda() if die else torch.empty(0, dtype=torch.int32, device="cuda")
    to, frm, left, gave_up = kd.replay_node_removals(d_die, n_nodes)
    if gave_up:
        assert style == 3 and n_die > 4000          # only the contrived long chains are handed to the host replay
        return
    assert left == want_left
    assert dict(zip(to.cpu().tolist(), frm.cpu().tolist())) == want_moves and to.numel() == len(want_moves)


# ---- the same two replays on 64-bit positions: what katome_dist_remove_dead_paths runs on rank 0 (dist_prune.hip) ----------
def _sparse_edge_removals(pos, mult, n):
    """_naive_edge_removals without the array: only touched positions are kept (n may be 2^33)"""
    occ = {}
    victims, dups = [], 0
    for d, c in sorted(zip(pos, mult), reverse=True):
        for r in range(c):
            if d < n:
                victims.append(occ.get(d, d))
                dups += r != 0
                occ[d] = occ.get(n - 1, n - 1)
                n -= 1
    return victims, {p: v for p, v in occ.items() if p < n and v != p}, n, dups


def _sparse_node_removals(die, n):
    occ, where = {}, {}
    for a, b in die:
        gone = sorted((v for v in (a, b) if v >= 0), key=lambda v: where.get(v, v), reverse=True)
        for v in gone:
            p = where.pop(v, v)
            last = occ.get(n - 1, n - 1)
            n -= 1
            if p < n:
                occ[p] = last
                where[last] = p
    return {p: v for p, v in occ.items() if p < n and v != p}, n


@pytest.mark.parametrize("seed", range(24))
def test_edge_removal_replay_beyond_32_bit_indices(seed):
    """BASELINE config 5 in full holds 1.1e10 edges: more than 2^32 indices.  The replay only touches the marked entries and
    the tail that disappears, so a graph of 2^33 + n edges is replayed here with marks at both ends of the index range
    (katome_dev_replay_edge_removals64) and compared with the literal swap_remove loop kept sparse.  seed % 3 == 0: small
    totals, where the 64-bit form must agree with the 32-bit one entry for entry."""
    from katome_amd import device as kd
    rng = np.random.default_rng(9000 + seed)
    small = seed % 3 == 0
    n_edges = int(rng.integers(1000, 30000)) if small else (1 << 33) + int(rng.integers(1, 1 << 20))
    tail = np.unique(n_edges - 1 - rng.integers(0, 6000, int(rng.integers(1, 4000))))
    low = np.unique(rng.integers(0, min(n_edges, 1 << 34) // 2, int(rng.integers(0, 3000))))
    pos = np.unique(np.concatenate([low, tail[tail >= 0]]))
    mult = rng.integers(1, (2, 5, 30)[seed % 3] + 1, len(pos))
    want_v, want_moves, want_left, want_dups = _sparse_edge_removals(pos.tolist(), mult.tolist(), n_edges)
    d_pos = torch.from_numpy(pos.astype(np.int64)).cuda()
    d_mult = torch.from_numpy(mult.astype(np.int64)).to(torch.int32).cuda()
    v, to, frm, left, dups = kd.replay_edge_removals64(d_pos, d_mult, n_edges)
    assert left == want_left and dups == want_dups
    assert v.cpu().tolist() == want_v
    assert dict(zip(to.cpu().tolist(), frm.cpu().tolist())) == want_moves and to.numel() == len(want_moves)
    if small:
        v32, to32, frm32, left32, dups32 = kd.replay_edge_removals(d_pos.to(torch.int32), d_mult, n_edges)
        assert (left32, dups32) == (left, dups) and v32.cpu().tolist() == want_v
        assert dict(zip(to32.cpu().tolist(), frm32.cpu().tolist())) == want_moves


@pytest.mark.parametrize("seed", range(24))
def test_node_removal_replay_beyond_32_bit_indices(seed):
    from katome_amd import device as kd
    rng = np.random.default_rng(9500 + seed)
    n_nodes = (1 << 33) + int(rng.integers(1, 1 << 20))
    n_die = int(rng.integers(1, 5000))
    tail = (n_nodes - 1 - rng.choice(8000, size=min(n_die, 8000) // 2 + 1, replace=False)).tolist()
    low = rng.choice(1 << 32, size=n_die // 2 + 1, replace=False).tolist()
    dying = [int(x) for x in rng.permutation(np.array(tail + low, dtype=np.int64))]
    die = []
    while dying:
        kind = rng.integers(0, 4)
        if kind == 0 and len(dying) >= 2:
            die.append((dying.pop(), dying.pop()))
        elif kind == 1:
            die.append((dying.pop(), -1))
        elif kind == 2:
            die.append((-1, dying.pop()))
        else:
            die.append((-1, -1))
    want_moves, want_left = _sparse_node_removals(die, n_nodes)
    d_die = torch.tensor(die, dtype=torch.int64).reshape(-1).cuda()
    to, frm, left, gave_up = kd.replay_node_removals64(d_die, n_nodes)
    assert not gave_up and left == want_left
    assert dict(zip(to.cpu().tolist(), frm.cpu().tolist())) == want_moves and to.numel() == len(want_moves)
