"""Oracle side of Shrinkable::shrink (shrinker.rs:38-209): the reference's own expectations -- the counts of
tests/shrinker.rs:33-36 on the fixtures and every traverse case of the in-file tests (shrinker.rs:291-488), which pin node
indices, degrees and merged labels."""
import json
import os

import pytest


@pytest.fixture(scope="module")
def pinned(golden_dir):
    with open(os.path.join(golden_dir, "pinned.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_pinned_shrink_counts(oracle, pinned, golden_dir, i):
    """tests/shrinker.rs:33-36,43-58: (62,61) -> (2,1), (5704,5612) -> (184,92), (14446,14213) -> (466,233)"""
    p = pinned["shrink"]
    g = oracle.build_files([os.path.join(golden_dir, pinned["fixtures"][i])], pinned["k"], False, stages="s")
    assert [g.n_nodes, g.n_edges] == p["counts"][i]
    # every merged edge spells a walk of the unshrunk graph: its k-mers are edges of the full build, its weight the first one's
    full = dict(oracle.build_files([os.path.join(golden_dir, pinned["fixtures"][i])], pinned["k"], False).multiset())
    k = pinned["k"]
    seen = 0
    for seq, w in zip(g.edge_seq, g.edge_weight):
        kmers = [seq[j:j + k] for j in range(len(seq) - k + 1)]
        assert all(km in full for km in kmers) and full[kmers[0]] == int(w)
        seen += len(kmers)
    assert seen == len(full)                      # every edge of the build is in exactly one merged edge


def _kat(golden_dir):
    with open(os.path.join(golden_dir, "shrinker_kat.json")) as f:
        return json.load(f)


def test_in_file_cases(oracle, golden_dir):
    kat = _kat(golden_dir)
    k = kat["k"]
    slots = [None] + ["A" * 37 + s for s in kat["slots"][1:]]
    for case in kat["cases"]:
        g = oracle.shrink_from_edges([tuple(e) for e in case["edges"]], slots, k)
        if case["nodes"] is not None:
            assert g.n_nodes == case["nodes"], case["name"]
        if case["n_edges"] is not None:
            assert g.n_edges == case["n_edges"], case["name"]
        src, dst = g.edge_src.tolist(), g.edge_dst.tolist()
        for node, indeg, outdeg in case["check_node"]:
            assert (dst.count(node), src.count(node)) == (indeg, outdeg), (case["name"], node)
        for a, b, suffix in case["check_edge"]:
            found = [g.edge_seq[e][37:] for e in range(g.n_edges) if (src[e], dst[e]) == (a, b)]
            assert found and found[0] == suffix, (case["name"], a, b, found)     # find_edge returns the first in list order
