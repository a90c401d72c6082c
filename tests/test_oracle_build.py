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


def test_bfcounter_restatement(oracle, golden_dir, tmp_path):
    """create_bfc (builder.rs:79-115): unpinned by the reference's tests; checked for self-consistency with the
    FASTQ path: the distinct k-mers of a FASTQ build, written as a BFCounter file, rebuild the same graph."""
    path = os.path.join(golden_dir, "data2.txt")
    ref = oracle.build_files([path], 31, False)
    bfc = tmp_path / "kmers.bfc"
    bfc.write_text("".join("%s\t%d\n" % (km, w) for km, w in ref.multiset()))
    g = oracle.build_bfc([str(bfc)], 31, False, 0)
    assert g.multiset() == ref.multiset() and g.n_nodes == ref.n_nodes
    assert g.read_bytes == 31 * ref.n_edges
    # threshold drops lines before anything is added (builder.rs:106-108)
    g2 = oracle.build_bfc([str(bfc)], 31, False, 2)
    assert g2.multiset() == [(km, w) for km, w in ref.multiset() if w >= 2]
    # reverse_complement adds the reverse complement of every line as a second edge (pt_graph.rs:321-324)
    g3 = oracle.build_bfc([str(bfc)], 31, True, 0)
    assert g3.n_edges == 2 * ref.n_edges
    with pytest.raises(oracle.OracleError) as e:
        bad = tmp_path / "bad.bfc"
        bad.write_text("ACGT\n")
        oracle.build_bfc([str(bad)], 4, False, 0)
    assert e.value.name == "E_PARSE"


@pytest.mark.parametrize("i", [0, 1, 2])
def test_pinned_remove_weak_edges(oracle, pinned, golden_dir, i):
    """Clean::remove_weak_edges for PtGraph (pruner.rs:84-93) against tests/pruner.rs's constants"""
    p = pinned["remove_weak_edges"]
    g = oracle.build_files([os.path.join(golden_dir, pinned["fixtures"][i])], pinned["k"], False,
                           remove_weak_edges=p["thresholds"][i])
    assert [g.n_nodes, g.n_edges] == p["counts"][i]
