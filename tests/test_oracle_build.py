"""Pin the oracle's restated PtGraph::create / HmGIR::create against the constants the
reference's integration tests hold (tests/golden/pinned.json cites each one)."""
import json
import os

import pytest


@pytest.fixture(scope="module")
def pinned(golden_dir):
    with open(os.path.join(golden_dir, "pinned.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_pinned_counts_and_stats(oracle, pinned, golden_dir, i):
    path = os.path.join(golden_dir, pinned["fixtures"][i])
    g = oracle.build_files([path], pinned["k"], pinned["reverse_complement"], file_type=1, with_gir=True)
    assert g.read_bytes == pinned["read_bytes"]["values"][i]
    assert [g.n_nodes, g.n_edges] == pinned["counts"]["values"][i]
    assert list(g.gir_counts) == pinned["counts"]["values"][i]          # HmGIR/HsGIR observable
    want = pinned["pt_graph_stats"]["values"][i]
    for key, val in want.items():
        got = g.stats[key]
        if isinstance(val, float):
            assert round(got * 100.0) / 100.0 == val, key   # stats/collections.rs:84-89
        else:
            assert got == val, key
    # SEQUENCES holds scratch + one slot per distinct edge (pt_graph.rs:176-191)
    assert g.n_sequences == g.n_edges + 1
    assert sorted(g.edge_slot.tolist()) == list(range(1, g.n_edges + 1))


def test_bad_path_fails(oracle, pinned, golden_dir):
    with pytest.raises(oracle.OracleError) as e:
        oracle.build_files([os.path.join(golden_dir, pinned["bad_path"]["name"])], 40)
    assert e.value.name == "E_PATH"


def test_error_semantics(oracle, golden_dir):
    # a read with a non-ACGT byte is skipped BEFORE the length check (builder.rs:155-158);
    # an accepted read shorter than k is fatal (pt_graph.rs:278)
    with pytest.raises(oracle.OracleError) as e:
        oracle.build_files([os.path.join(golden_dir, "data_too_short_read.txt")], 40)
    assert e.value.name == "E_SHORT_READ"
    g = oracle.build_files([os.path.join(golden_dir, "data_too_short_read.txt")], 7)
    assert (g.read_bytes, g.n_edges) == (7, 1)
    with pytest.raises(oracle.OracleError) as e:
        oracle.build_files([golden_dir], 40)
    assert e.value.name == "E_IS_DIR"


def test_edge_labels_are_kmers_of_reads(oracle, golden_dir):
    """every label decodes (decompress_edge) to a k-mer of an accepted read, weights sum to
    the number of windows"""
    path = os.path.join(golden_dir, "data2.txt")
    reads = oracle.scan_files([path])
    assert (reads["n_records"], reads["n_accepted"], reads["read_bytes"]) == (125, 92, 9200)
    for k, rc in ((40, False), (31, True)):
        g = oracle.build_files([path], k, rc)
        windows = {}
        comp = bytes.maketrans(b"ACGT", b"TGCA")
        for r in range(reads["n_accepted"]):
            s = bytes(reads["seq"][reads["off"][r]:reads["off"][r + 1]])
            for strand in ((s, s.translate(comp)[::-1]) if rc else (s,)):
                for w in range(len(strand) - k + 1):
                    km = strand[w:w + k].decode()
                    windows[km] = windows.get(km, 0) + 1
        assert g.multiset() == sorted(windows.items())


def test_multifile_is_concatenation(oracle, golden_dir):
    p = [os.path.join(golden_dir, f) for f in ("data1.txt", "data3.txt")]
    g = oracle.build_files(p, 40)
    a, b = oracle.build_files(p[:1], 40), oracle.build_files(p[1:], 40)
    assert g.read_bytes == a.read_bytes + b.read_bytes
    merged = {}
    for km, w in a.multiset() + b.multiset():
        merged[km] = merged.get(km, 0) + w
    assert g.multiset() == sorted(merged.items())


def test_derived_goldens(oracle, golden_dir):
    """Derived (NOT reference-pinned) goldens, produced by the oracle after it passed the pins."""
    with open(os.path.join(golden_dir, "derived.json")) as f:
        derived = json.load(f)
    for case in derived["cases"]:
        g = oracle.build_files([os.path.join(golden_dir, case["fixture"])], case["k"], case["rc"])
        assert [g.n_nodes, g.n_edges] == case["counts"]
        assert int(g.edge_weight.astype("uint64").sum()) == case["weight_sum"]
