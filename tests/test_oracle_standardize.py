"""Oracle side of Standardizable (standardizer.rs:41-128): every in-file case of the reference (standardizer.rs:130-313),
which pins edge weights by index; nothing else of it is pinned (tests/standardizer.rs is commented out)."""
import json
import os

import pytest


def _kat(golden_dir):
    with open(os.path.join(golden_dir, "standardizer_kat.json")) as f:
        return json.load(f)


def test_in_file_cases(oracle, golden_dir):
    for case in _kat(golden_dir)["cases"]:
        oracle.set_genome_length(case.get("genome_length", 0))
        g = oracle.run_from_edges(case["n_nodes"], [tuple(e) for e in case["edges"]], case["stages"], case.get("threshold", 0),
                                  case.get("k", 40))
        assert g.n_edges == case.get("n_edges", len(case["edges"])), case["name"]
        for i, w in enumerate(case["weights"]):
            if w is not None:
                assert int(g.edge_weight[i]) == w, (case["name"], i)


def test_empty_graph(oracle):
    """standardizer.rs:145-153"""
    g = oracle.run_from_edges(0, [], "c")
    assert (g.n_nodes, g.n_edges) == (0, 0)


@pytest.mark.parametrize("k,rc", [(21, True), (8, True), (12, False)])
def test_contig_means_on_a_build(oracle, golden_dir, k, rc):
    """standardize_contigs only changes weights: sums per contig are kept up to rounding, the graph is untouched, and a
    second application changes nothing"""
    path = [os.path.join(golden_dir, "data3.txt")]
    before = oracle.build_files(path, k, rc)
    once = oracle.build_files(path, k, rc, stages="c")
    twice = oracle.build_files(path, k, rc, stages="cc")
    assert (once.edge_src.tolist(), once.edge_dst.tolist()) == (before.edge_src.tolist(), before.edge_dst.tolist())
    assert once.edge_weight.tolist() == twice.edge_weight.tolist()
    assert abs(int(once.edge_weight.astype("int64").sum()) - int(before.edge_weight.astype("int64").sum())) <= before.n_edges
