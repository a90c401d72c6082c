"""Shrinkable::shrink (shrinker.rs:165-209) through the C ABI against the oracle's literal restatement: the merged edges
as a multiset of (sequence, weight) with their end vertices -- the numbering of the result is the library's own."""
import json
import os

import numpy as np
import pytest

from helpers import int_to_kmer, pack_reads_ascii

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _contigs(dc, k):
    """[(sequence, weight)] sorted, after checking labels against end vertices and path lengths"""
    seqs = dc.sequences()
    w = dc.edge_weight.cpu().numpy().view(np.uint32).tolist()
    nk = dc.node_key.cpu().numpy().view(np.uint64)
    names = [int_to_kmer(int(r[0]) if dc.key_words == 1 else (int(r[0]) << 64) | int(r[1]), k - 1) for r in nk]
    src, dst = dc.edge_src.cpu().tolist(), dc.edge_dst.cpu().tolist()
    kmers = dc.edge_kmers.cpu().tolist()
    assert len(set(names)) == len(names)
    for i, s in enumerate(seqs):
        assert len(s) == k + kmers[i] - 1
        assert names[src[i]] == s[:k - 1] and names[dst[i]] == s[-(k - 1):]
    assert set(src) | set(dst) == set(range(dc.n_nodes))           # no vertex without an edge is left
    return sorted(zip(seqs, w))


def _build(kd, packed, n, L, k, rc, first_seen=False):
    b = kd.Builder(k, rc, first_seen_order=first_seen)
    span = b.tile_span(L)
    if span > 1:
        b.insert_tiles(b.extract_tiles(packed, n, L, span), span)
    else:
        b.insert(b.extract_fixed(packed, n, L))
    b.finalize()
    return b


@pytest.mark.parametrize("i", [0, 1, 2])
def test_reference_pinned_counts(oracle, golden_dir, i):
    """tests/shrinker.rs:33-36: (2,1), (184,92), (466,233) -- and the same merged edges as the oracle"""
    from katome_amd import device as kd
    from katome_amd.build import ingest_files, InputFileType
    pinned = json.load(open(os.path.join(golden_dir, "pinned.json")))
    path, k = os.path.join(golden_dir, pinned["fixtures"][i]), pinned["k"]
    r = ingest_files([path], InputFileType.Fastq, k)
    packed = torch.from_numpy(r["packed"].copy()).cuda()
    for first_seen in (False, True):
        b = _build(kd, packed, r["n_reads"], r["fixed_len"], k, False, first_seen)
        dc = b.shrink()
        assert [dc.n_nodes, dc.n_edges] == pinned["shrink"]["counts"][i]
        assert _contigs(dc, k) == oracle.build_files([path], k, False, stages="s").contigs()
        b.close()


def _reaches_everything_from_inputs(g):
    """True when every vertex is reachable from a vertex without incoming edges -- the part of a graph on which the
    reference's result does not depend on its traversal order (shrink)"""
    n = g.n_nodes
    out = [[] for _ in range(n)]
    indeg = [0] * n
    for s, d in zip(g.edge_src.tolist(), g.edge_dst.tolist()):
        out[s].append(d)
        indeg[d] += 1
    seen = [indeg[v] == 0 for v in range(n)]
    stack = [v for v in range(n) if seen[v]]
    while stack:
        v = stack.pop()
        for d in out[v]:
            if not seen[d]:
                seen[d] = True
                stack.append(d)
    return all(seen)


@pytest.mark.parametrize("k,rc,n,L,glen,err", [(31, True, 4000, 100, 30000, 5e-3), (21, False, 3000, 80, 20000, 1e-2),
                                               (40, True, 2500, 103, 20000, 5e-3), (15, True, 3000, 60, 8000, 1e-2),
                                               (33, False, 2000, 150, 25000, 0.0)])
def test_synthetic_reads_against_oracle(oracle, k, rc, n, L, glen, err):
    from katome_amd import device as kd
    ascii_reads = oracle.synth_reads(0, n, L, glen, err, 0)
    packed = torch.from_numpy(pack_reads_ascii(ascii_reads).reshape(-1).copy()).cuda()
    full = oracle.build_ascii(ascii_reads, k, rc)
    assert _reaches_everything_from_inputs(full)
    want = oracle.build_ascii(ascii_reads, k, rc, stages="s")
    b = _build(kd, packed, n, L, k, rc)
    dc = b.shrink()
    assert (dc.n_nodes, dc.n_edges) == (want.n_nodes, want.n_edges)
    assert _contigs(dc, k) == want.contigs()
    assert dc.n_edges < full.n_edges
    b.close()
    # the assembler's order: remove_dead_paths, then shrink (asm/basic_assembler.rs:58-65)
    b = _build(kd, packed, n, L, k, rc, first_seen=True)
    b.remove_dead_paths()
    pruned = oracle.build_ascii(ascii_reads, k, rc, remove_dead_paths=True)
    if pruned.n_edges and _reaches_everything_from_inputs(pruned):
        assert _contigs(b.shrink(), k) == oracle.build_ascii(ascii_reads, k, rc, stages="ds").contigs()
    b.close()


@pytest.mark.parametrize("k,rc", [(11, False), (16, True), (31, True)])
def test_cycle_of_inner_vertices(oracle, k, rc):
    """reads off a circular sequence and nothing else: one cycle without any vertex to start from; it becomes a
    self-loop spelling the circle once (where it is cut is traversal business: compared up to rotation)"""
    from katome_amd import device as kd
    rng = np.random.default_rng(k)
    circle = "".join("ACGT"[c] for c in rng.integers(0, 4, 300))
    L = 60
    reads = np.array([[ord(c) for c in (circle + circle)[s:s + L]] for s in range(0, 300, 7)], dtype=np.uint8)
    packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
    b = _build(kd, packed, len(reads), L, k, rc)
    dc = b.shrink()
    got = _contigs(dc, k)
    want = oracle.build_ascii(reads, k, rc, stages="s").contigs()
    assert len(got) == len(want) == (2 if rc else 1)
    assert all(len(s) == 300 + k - 1 for s, _ in got + want)

    def canon(seq):                      # the circle a self-loop label spells, rotated to its smallest form
        c = seq[:len(seq) - (k - 1)]
        return min(c[i:] + c[:i] for i in range(len(c)))
    assert sorted(canon(s) for s, _ in got) == sorted(canon(t) for t, _ in want)
    assert dc.edge_src.cpu().tolist() == dc.edge_dst.cpu().tolist()
    b.close()


def _reachable_names(g, k):
    """(k-1)-mers of the vertices of the unshrunk oracle graph that a vertex without incoming edges reaches"""
    n = g.n_nodes
    out = [[] for _ in range(n)]
    indeg = [0] * n
    name = [None] * n
    for s, d, km in zip(g.edge_src.tolist(), g.edge_dst.tolist(), g.kmer_strings()):
        out[s].append(d)
        indeg[d] += 1
        name[s], name[d] = km[:-1], km[1:]
    seen = [indeg[v] == 0 for v in range(n)]
    stack = [v for v in range(n) if seen[v]]
    while stack:
        v = stack.pop()
        for d in out[v]:
            if not seen[d]:
                seen[d] = True
                stack.append(d)
    return {name[v] for v in range(n) if seen[v]}


@pytest.mark.parametrize("seed", range(16))
def test_tangled_graphs_where_the_traversal_order_cannot_matter(oracle, seed):
    """small k and noisy reads: branches, self-loops, two-cycles, repeats.  Compared on the merged edges that start at a
    vertex some in-degree-0 vertex reaches; elsewhere the reference's result depends on where its traversal happens to
    enter and only the totals are sanity-checked"""
    from katome_amd import device as kd
    k = [5, 6, 7, 8, 9, 11, 13, 17][seed % 8]
    rng = np.random.default_rng(50 + seed)
    genome = rng.integers(0, 4, 200 + 60 * seed)
    L = k + 6 + seed % 9
    reads = np.zeros((60 + 25 * seed, L), np.uint8)
    for i in range(len(reads)):
        s0 = rng.integers(0, len(genome) - L + 1)
        r = genome[s0:s0 + L].copy()
        m = rng.random(L) < 0.03
        r[m] = rng.integers(0, 4, int(m.sum()))
        reads[i] = np.frombuffer(b"ACGT", np.uint8)[r]
    packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
    for rc in (False, True):
        full = oracle.build_ascii(reads, k, rc)
        ok = _reachable_names(full, k)
        want = [c for c in oracle.build_ascii(reads, k, rc, stages="s").contigs() if c[0][:k - 1] in ok]
        b = _build(kd, packed, len(reads), L, k, rc)
        dc = b.shrink()
        got_all = _contigs(dc, k)
        assert [c for c in got_all if c[0][:k - 1] in ok] == want
        assert want or k <= 6                 # (at k = 5, 6 nearly every (k-1)-mer has a predecessor: nothing to start from)
        assert sum(dc.edge_kmers.cpu().tolist()) == full.n_edges          # every k-mer of the build is in exactly one merged edge
        b.close()


def _same_as_reference(dc, want, k):
    """the shrunk graph index for index: end points, weights, sequences, in petgraph's numbering after shrink
    (shrinker.rs:165-209: swap_removes and add_edges, then remove_single_vertices)"""
    assert (dc.n_nodes, dc.n_edges) == (want.n_nodes, want.n_edges)
    assert dc.edge_src.cpu().tolist() == want.edge_src.tolist() and dc.edge_dst.cpu().tolist() == want.edge_dst.tolist()
    assert dc.edge_weight.cpu().numpy().view(np.uint32).tolist() == want.edge_weight.tolist()
    assert dc.sequences() == want.edge_seq
    _contigs(dc, k)                      # (labels against end vertices' keys and path lengths: the node keys follow the numbering too)


@pytest.mark.parametrize("seed", range(16))
def test_exact_shrink_on_tangled_graphs_index_for_index(oracle, seed):
    """the tangles of the test above -- branches, self-loops, two-cycles, parts no vertex without incoming edges reaches -- through
    the EXACT form: where the reference's traversal cuts, which restarts it makes (and which its offset makes it skip), and the
    indices its swap_removes leave; every array against the oracle's literal petgraph"""
    from katome_amd import device as kd
    k = [5, 6, 7, 8, 9, 11, 13, 17][seed % 8]
    rng = np.random.default_rng(50 + seed)
    genome = rng.integers(0, 4, 200 + 60 * seed)
    L = k + 6 + seed % 9
    reads = np.zeros((60 + 25 * seed, L), np.uint8)
    for i in range(len(reads)):
        s0 = rng.integers(0, len(genome) - L + 1)
        r = genome[s0:s0 + L].copy()
        m = rng.random(L) < 0.03
        r[m] = rng.integers(0, 4, int(m.sum()))
        reads[i] = np.frombuffer(b"ACGT", np.uint8)[r]
    packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
    for rc in (False, True):
        b = _build(kd, packed, len(reads), L, k, rc, first_seen=True)
        _same_as_reference(b.shrink("exact"), oracle.build_ascii(reads, k, rc, stages="s"), k)
        # the fast form on the same graph: the same k-mers in its merged edges, its own numbering
        assert sum(b.shrink("fast").edge_kmers.cpu().tolist()) == oracle.build_ascii(reads, k, rc).n_edges
        b.close()


@pytest.mark.parametrize("k,rc", [(11, False), (16, True), (31, True)])
def test_exact_shrink_of_cycles_index_for_index(oracle, k, rc):
    """reads off circular sequences only: no vertex to start from, the traversal's restarts decide where each circle is cut"""
    from katome_amd import device as kd
    rng = np.random.default_rng(k)
    reads = []
    for n in (300, 170, 90):
        circle = "".join("ACGT"[c] for c in rng.integers(0, 4, n))
        reads += [[ord(c) for c in (circle * 3)[s:s + 60]] for s in range(0, n, 7)]
    order = rng.permutation(len(reads))
    reads = np.array([reads[i] for i in order], dtype=np.uint8)
    packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
    b = _build(kd, packed, len(reads), 60, k, rc, first_seen=True)
    _same_as_reference(b.shrink(), oracle.build_ascii(reads, k, rc, stages="s"), k)       # (auto = exact on this builder)
    b.close()


@pytest.mark.parametrize("k,rc,n,L,glen,err", [(31, True, 4000, 100, 30000, 5e-3), (21, False, 3000, 80, 20000, 1e-2),
                                               (40, True, 2500, 103, 20000, 5e-3), (15, True, 3000, 60, 8000, 1e-2)])
def test_exact_shrink_after_pruning_index_for_index(oracle, k, rc, n, L, glen, err):
    """the assembler's order (asm/basic_assembler.rs:58-65; collapser.rs:31): remove_dead_paths, whose swap_removes leave the edges
    out of age order (the adjacency lists still follow the ages), then shrink -- and shrink alone"""
    from katome_amd import device as kd
    ascii_reads = oracle.synth_reads(0, n, L, glen, err, 0)
    packed = torch.from_numpy(pack_reads_ascii(ascii_reads).reshape(-1).copy()).cuda()
    b = _build(kd, packed, n, L, k, rc, first_seen=True)
    _same_as_reference(b.shrink("exact"), oracle.build_ascii(ascii_reads, k, rc, stages="s"), k)
    b.remove_dead_paths()
    want = oracle.build_ascii(ascii_reads, k, rc, stages="ds")
    if want.n_edges:
        _same_as_reference(b.shrink("exact"), want, k)
        assert b.last_shrink_host_ms > 0
    b.close()


@pytest.mark.parametrize("i", [0, 1, 2])
def test_exact_shrink_of_the_reference_fixtures(oracle, golden_dir, i):
    """tests/shrinker.rs:33-36's counts again, and every array of the three fixtures' shrunk graphs"""
    from katome_amd import device as kd
    from katome_amd.build import ingest_files, InputFileType
    pinned = json.load(open(os.path.join(golden_dir, "pinned.json")))
    path, k = os.path.join(golden_dir, pinned["fixtures"][i]), pinned["k"]
    r = ingest_files([path], InputFileType.Fastq, k)
    packed = torch.from_numpy(r["packed"].copy()).cuda()
    for rc in (False, True):
        b = _build(kd, packed, r["n_reads"], r["fixed_len"], k, rc, True)
        dc = b.shrink("exact")
        if not rc:
            assert [dc.n_nodes, dc.n_edges] == pinned["shrink"]["counts"][i]
        _same_as_reference(dc, oracle.build_files([path], k, rc, stages="s"), k)
        b.close()


@pytest.mark.parametrize("name,k,rc,prune", [("data2.txt", 40, False, False), ("data3.txt", 40, True, True), ("data3.txt", 21, True, True),
                                             ("data2.txt", 63, True, False)])
def test_host_entry(oracle, golden_dir, name, k, rc, prune):
    """katome_shrink_files: ingest, build, (first pruning,) shrink, host arrays"""
    from katome_amd.build import GpuContigs, InputFileType, set_global_k_sizes
    path = os.path.join(golden_dir, name)
    set_global_k_sizes(k)
    c, rb = GpuContigs.create([path], InputFileType.Fastq, rc, 0, first_seen_order=prune, remove_dead_paths=prune)
    want = oracle.build_files([path], k, rc, stages="ds" if prune else "s")
    assert rb == want.read_bytes and (c.n_nodes, c.n_edges) == (want.n_nodes, want.n_edges)
    assert c.contigs() == want.contigs()
    assert all(len(s) == k + int(m) - 1 for s, m in zip(c.edge_seq, c.edge_kmers))


def test_before_finalize_is_an_error():
    from katome_amd import device as kd
    from katome_amd.build import KatomePanic
    b = kd.Builder(21, True)
    with pytest.raises(KatomePanic):
        b.shrink()
    b.close()
