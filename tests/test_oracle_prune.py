"""Oracle side of Prunable::remove_dead_paths (pruner.rs:36-82): the reference's pinned counts, structural checks of
the petgraph swap_remove restatement, and the decomposition prune.hip uses (tests/prune_model.py) against it."""
import os

import numpy as np
import pytest

import json

import prune_model


@pytest.fixture(scope="module")
def pinned(golden_dir):
    with open(os.path.join(golden_dir, "pinned.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_pinned_remove_dead_paths(oracle, pinned, golden_dir, i):
    """tests/pruner.rs:204-216 with the expected counts of tests/pruner.rs:37-169"""
    g = oracle.build_files([os.path.join(golden_dir, pinned["fixtures"][i])], pinned["k"], False, remove_dead_paths=True)
    assert [g.n_nodes, g.n_edges] == pinned["remove_dead_paths"]["counts"][i]
    assert g.stats["node_count"] == 0 and g.stats["edge_count"] == 0


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


@pytest.mark.parametrize("seed", range(12))
def test_pruned_graph_is_a_consistent_subgraph(oracle, seed):
    """after all the swap_removes every surviving edge still carries its own weight and label, endpoints that share
    an id share a (k-1)-mer, ids are dense, and no node is left without an edge"""
    k = [4, 5, 6, 8, 11, 17][seed % 6]
    reads = _random_reads(seed, 60 + 30 * seed, k + 4 + seed % 5, 80 + 40 * seed, 0.05)
    for rc in (False, True):
        full = oracle.build_ascii(reads, k, rc)
        g = oracle.build_ascii(reads, k, rc, remove_dead_paths=True)
        weight_of = dict(full.multiset())
        kmers = g.kmer_strings()
        assert len(set(kmers)) == len(kmers)
        node_seq = {}
        for e, km in enumerate(kmers):
            assert int(g.edge_weight[e]) == weight_of[km]
            for node, seq in ((int(g.edge_src[e]), km[:-1]), (int(g.edge_dst[e]), km[1:])):
                assert node_seq.setdefault(node, seq) == seq
        assert sorted(node_seq) == list(range(g.n_nodes))
        assert len(set(node_seq.values())) == g.n_nodes


@pytest.mark.parametrize("seed", range(30))
def test_decomposition_matches_petgraph_replay(oracle, seed):
    """walk summary + marked indices from the top + the two replays == the literal sequential algorithm"""
    k = [4, 5, 6, 8, 11][seed % 5]
    reads = _random_reads(100 + seed, 30 + seed * 3, k + 3 + seed % 7, 60 + seed * 10, 0.05)
    for rc in (False, True):
        g0 = oracle.build_ascii(reads, k, rc)
        g1 = oracle.build_ascii(reads, k, rc, remove_dead_paths=True)
        passes = oracle.last_prune_passes()
        src, dst, orig, _, st = prune_model.remove_dead_paths(g0.edge_src, g0.edge_dst, k)
        assert src == [int(x) for x in g1.edge_src] and dst == [int(x) for x in g1.edge_dst]
        assert np.array_equal(g1.edge_label, g0.edge_label[orig].reshape(g1.edge_label.shape))
        assert np.array_equal(g1.edge_weight, g0.edge_weight[orig])
        assert st["passes"] == passes


def test_model_reaches_the_quirks(oracle):
    dup = loops = passes = 0
    for seed in range(30):
        k = [4, 5, 6, 8, 11][seed % 5]
        reads = _random_reads(100 + seed, 30 + seed * 3, k + 3 + seed % 7, 60 + seed * 10, 0.05)
        g0 = oracle.build_ascii(reads, k, True)
        st = prune_model.remove_dead_paths(g0.edge_src, g0.edge_dst, k)[4]
        dup += st["removed_by_duplicates"]
        loops += st["self_loops"]
        passes = max(passes, st["passes"])
    assert dup > 0 and passes > 2


# the in-file tests of pruner.rs (259-393): hand-built graphs, counts after Clean's two operations
PRUNER_KAT = [
    # (name, lines, n_nodes, edges (src, dst, weight), stages, threshold, expected (nodes, edges))
    ("prunes_single_graph", "266-274", 0, [], "w", 10, (0, 0)),
    ("prunes_single_weak_edge", "289-298", 3, [(0, 1, 100), (1, 2, 1)], "w", 10, (2, 1)),
    ("prunes_single_weak_edge_and_no_nodes", "300-311", 4, [(0, 1, 100), (1, 2, 1), (2, 3, 100)], "w", 10, (4, 2)),
    ("prunes_strong_edges", "313-322", 3, [(0, 1, 100), (1, 2, 100)], "w", 10, (3, 2)),
    ("prunes_cycle", "324-334", 3, [(0, 1, 1), (1, 2, 1), (2, 0, 1)], "w", 10, (0, 0)),
    ("doesnt_remove_vertices", "339-348", 3, [(0, 1, 100), (1, 2, 1)], "v", 0, (3, 2)),
    ("removes_one_vertex", "350-358", 3, [(0, 1, 100)], "v", 0, (2, 1)),
    ("removes_two_vertices", "360-368", 3, [(0, 0, 100)], "v", 0, (1, 1)),
    ("removes_all_vertices", "370-377", 3, [], "v", 0, (0, 0)),
]


@pytest.mark.parametrize("name,lines,n_nodes,edges,stages,thr,want", PRUNER_KAT, ids=[c[0] for c in PRUNER_KAT])
def test_pruner_in_file_cases(oracle, name, lines, n_nodes, edges, stages, thr, want):
    g = oracle.run_from_edges(n_nodes, edges, stages, thr)
    assert (g.n_nodes, g.n_edges) == want
    if name == "removes_two_vertices":          # the vertex that keeps its self-loop is re-labelled 0 by the swap_removes
        assert g.edge_src.tolist() == g.edge_dst.tolist() == [0]
