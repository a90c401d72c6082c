"""Oracle side of Collapsable::collapse (collapser.rs:29-273): the reference's own expectations -- the contig counts of
tests/collapser.rs:32 on the fixtures and every in-file case (collapser.rs:333-467), which pin the contig strings and
their order.  The oracle's collapse only exists to check, end to end, that the graph the GPU stages hand over yields the
reference's contigs (tests/test_gpu_prune.py)."""
import json
import os

import pytest


@pytest.fixture(scope="module")
def pinned(golden_dir):
    with open(os.path.join(golden_dir, "pinned.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_pinned_contig_counts(oracle, pinned, golden_dir, i):
    """tests/collapser.rs:32: [2, 92, 233] (data1: one 61-edge path of weight 2 is walked twice)"""
    g = oracle.build_files([os.path.join(golden_dir, pinned["fixtures"][i])], pinned["k"], False, stages="C")
    assert len(g.collapsed) == pinned["collapse"]["contigs"][i]
    assert (g.n_nodes, g.n_edges) == (0, 0)
    assert all(len(c) == 100 for c in g.collapsed)          # every contig spells a whole 100-bp read of the fixture


def test_in_file_cases(oracle, golden_dir):
    with open(os.path.join(golden_dir, "collapser_kat.json")) as f:
        kat = json.load(f)
    name, second, k = kat["name"], kat["second"], kat["k"]
    slots = [None, name[:k], name[1:k + 1], name[2:k + 2], name[3:k + 3], second[:k]]
    for case in kat["cases"]:
        g = oracle.run_from_edges(case["n_nodes"], [tuple(e) for e in case["edges"]], "C", 0, k, slots)
        assert g.collapsed == case["contigs"], case["name"]
