"""No-GPU checks of the C-ABI library: it loads, exports every symbol include/katome_gpu.h declares,
its host ingest agrees with the oracle, error statuses mirror the reference's panics, and the
device entry points fail loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import katome_amd
from katome_amd import _lib
from katome_amd.build import GpuGraph, InputFileType, KatomePanic, ingest_files, make_settings, set_global_k_sizes

from helpers import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "katome_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(katome_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.lib_path())
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == names          # the Python binding covers the whole header
    assert katome_amd.lib().katome_abi_version() == 2


def test_record_words():
    L = katome_amd.lib()
    assert [L.katome_record_words(k) for k in (3, 31, 32, 40, 63)] == [1, 1, 2, 2, 2]


@pytest.mark.parametrize("name,k", [("data1.txt", 40), ("data2.txt", 40), ("data3.txt", 31), ("data2.txt", 63)])
def test_ingest_matches_oracle(oracle, golden_dir, name, k):
    path = os.path.join(golden_dir, name)
    got = ingest_files([path], InputFileType.Fastq, k)
    want = oracle.scan_files([path], file_type=1)
    assert got["n_records"] == want["n_records"]
    assert got["n_reads"] == want["n_accepted"]
    assert got["read_bytes"] == want["read_bytes"]
    assert got["fixed_len"] == 100
    oracle.set_k(k)
    for r in range(got["n_reads"]):
        seq = bytes(want["seq"][want["off"][r]:want["off"][r + 1]])
        assert bytes(got["packed"][got["byte_off"][r]:got["byte_off"][r + 1]]) == oracle.compress_node(seq)
        assert got["len"][r] == len(seq)
    assert got["total_windows"] == sum(int(l) - k + 1 for l in got["len"])


def test_ingest_variable_length_and_fasta(tmp_path, oracle):
    fq = tmp_path / "v.fq"
    fq.write_text("@a\nACGTACGTAC\n+\nIIIIIIIIII\n@b\nACGTNACGT\n+\nIIIIIIIII\n@c\nTTTTGGGGCCCCAAAA  \n+\nIIIIIIIIIIIIIIII\n")
    got = ingest_files([str(fq)], InputFileType.Fastq, 5)
    assert (got["n_records"], got["n_reads"], got["read_bytes"], got["fixed_len"]) == (3, 2, 26, 0)
    want = oracle.scan_files([str(fq)], file_type=1)
    assert got["read_bytes"] == want["read_bytes"] and got["n_reads"] == want["n_accepted"]
    fa = tmp_path / "v.fa"
    fa.write_text(">x desc\nACGTAC\nGGTT\n>y\nACNT\n>z\nCCCCCCC\n")
    got = ingest_files([str(fa)], InputFileType.Fasta, 4)
    want = oracle.scan_files([str(fa)], file_type=0)
    assert (got["n_records"], got["n_reads"], got["read_bytes"]) == (3, 2, 17)
    assert (want["n_records"], want["n_accepted"], want["read_bytes"]) == (3, 2, 17)
    oracle.set_k(4)
    assert bytes(got["packed"][:3]) == oracle.compress_node(b"ACGTACGGTT")


def test_error_statuses_mirror_reference_panics(golden_dir, tmp_path):
    def status(paths, k=40, ft=InputFileType.Fastq):
        with pytest.raises(KatomePanic) as e:
            ingest_files(paths, ft, k)
        return e.value.name, e.value.message
    # tests/build.rs:33,129-139 -- a path that does not exist
    assert status([os.path.join(golden_dir, "data_too_short_reads")])[0] == "E_PATH"
    assert status([golden_dir])[0] == "E_IS_DIR"                                    # builder.rs:67
    name, msg = status([os.path.join(golden_dir, "data_too_short_read.txt")])        # pt_graph.rs:278
    assert (name, msg) == ("E_SHORT_READ", "Read is too short!")
    bad = tmp_path / "bad.fq"
    bad.write_text("ACGT\nACGT\n+\nIIII\n")
    assert status([str(bad)], k=3)[0] == "E_PARSE"                                  # builder.rs:153
    trunc = tmp_path / "trunc.fq"
    trunc.write_text("@a\nACGT\n+\n")
    assert status([str(trunc)], k=3)[0] == "E_PARSE"
    assert status([os.path.join(golden_dir, "data1.txt")], k=1)[0] == "E_ARG"       # prelude.rs:35
    assert status([os.path.join(golden_dir, "data1.txt")], k=64)[0] == "E_UNSUPPORTED"   # fine for the reference, two-word keys end at k = 63
    # errors are found in input order: the bad path wins over the good one before it (builder.rs:46)
    assert status([os.path.join(golden_dir, "data1.txt"), os.path.join(golden_dir, "nope")])[0] == "E_PATH"


def test_no_gpu_means_loud_failure(golden_dir):
    """On a box without a GPU the build must fail with E_DEVICE -- never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    set_global_k_sizes(40)
    with pytest.raises(KatomePanic) as e:
        GpuGraph.create([os.path.join(golden_dir, "data1.txt")], InputFileType.Fastq, False, 0)
    assert e.value.name == "E_DEVICE"
    s = make_settings(31)
    h = C.c_void_p()
    assert katome_amd.lib().katome_builder_create(C.byref(s), C.byref(h)) == -8
    packed = np.zeros(38, np.uint8)
    with pytest.raises(KatomePanic) as e:
        GpuGraph.create_from_packed(packed, 1, 150, k=31)
    assert e.value.name == "E_DEVICE"


def test_product_never_touches_the_oracle():
    """nothing under katome_amd/ may import, link or call oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "katome_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "katome_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def test_bfcounter_ingest(tmp_path):
    """`kmer<TAB>weight` lines (builder.rs:79-115): threshold, byte count, error statuses -- host side"""
    f = tmp_path / "a.bfc"
    f.write_text("ACGTA\t3\nCCCCC\t1\r\nGGGTT\t+7\textra\n")
    from katome_amd.build import make_settings
    import ctypes as C
    s = make_settings(5, InputFileType.BFCounter, min_weight=2)
    rp = C.POINTER(_lib.Reads)()
    arr = (C.c_char_p * 1)(os.fsencode(str(f)))
    assert katome_amd.lib().katome_ingest_files(C.byref(s), arr, 1, C.byref(rp)) == 0
    r = rp.contents
    assert (r.n_records, r.n_reads, r.read_bytes, r.fixed_len) == (3, 2, 10, 5)
    katome_amd.lib().katome_reads_free(rp)
    for text, k, want in (("ACGTA 3\n", 5, "E_PARSE"), ("ACGTA\tx\n", 5, "E_PARSE"), ("ACGTA\t99999999999\n", 5, "E_PARSE"),
                          ("ACG\t1\n", 5, "E_SHORT_READ"), ("ACGTAC\t1\n", 5, "E_ARG"), ("ACNTA\t1\n", 5, "E_PARSE")):
        f.write_text(text)
        with pytest.raises(KatomePanic) as e:
            ingest_files([str(f)], InputFileType.BFCounter, k)
        assert e.value.name == want, text


def test_parallel_ingest_equals_sequential(tmp_path, monkeypatch, oracle):
    """the multi-threaded scan (file cut at record starts) gives byte-identical output and the same first error"""
    rng = np.random.default_rng(3)
    fq_lines, fa_lines, bfc_lines = [], [], []
    for i in range(3000):
        n = int(rng.integers(20, 160))
        s = "".join("ACGT"[c] for c in rng.integers(0, 4, n))
        if i % 11 == 0:
            s = s[:3] + "N" + s[4:]
        q = "".join(chr(int(c)) for c in rng.integers(33, 74, n))      # qualities may start with '@' or '+'
        fq_lines += ["@r%d" % i, s, "+", "@" + q[1:] if i % 5 == 0 else q]
        fa_lines += [">r%d" % i] + [s[j:j + 37] for j in range(0, n, 37)]
        bfc_lines.append("%s\t%d" % ("".join("ACGT"[c] for c in rng.integers(0, 4, 21)), int(rng.integers(1, 9))))
    files = {"fq": (tmp_path / "p.fq", "\n".join(fq_lines) + "\n", InputFileType.Fastq, 15),
             "fa": (tmp_path / "p.fa", "\n".join(fa_lines) + "\n", InputFileType.Fasta, 15),
             "bfc": (tmp_path / "p.bfc", "\n".join(bfc_lines) + "\n", InputFileType.BFCounter, 21)}
    for name, (path, text, ft, k) in files.items():
        path.write_text(text)
        monkeypatch.setenv("KATOME_INGEST_THREADS", "1")
        seq = ingest_files([str(path)], ft, k)
        monkeypatch.setenv("KATOME_INGEST_MIN_CHUNK", "1000")
        for threads in ("2", "5", "32"):
            monkeypatch.setenv("KATOME_INGEST_THREADS", threads)
            par = ingest_files([str(path)], ft, k)
            for key in ("n_records", "n_reads", "read_bytes", "packed_bytes", "total_windows", "fixed_len"):
                assert par[key] == seq[key], (name, threads, key)
            for key in ("packed", "byte_off", "len"):
                assert np.array_equal(par[key], seq[key]), (name, threads, key)
        monkeypatch.delenv("KATOME_INGEST_MIN_CHUNK")
    # first error in file order wins, whichever thread meets it: a short read late in the file, a broken record later still
    path, text, ft, k = files["fq"]
    lines = text.split("\n")
    lines[4 * 2000 + 1] = "ACGTACG"                      # record 2000: accepted but shorter than k
    lines[4 * 2500] = "broken header"                    # record 2500: parse error, never reached
    path.write_text("\n".join(lines))
    monkeypatch.setenv("KATOME_INGEST_MIN_CHUNK", "1000")
    for threads in ("1", "7"):
        monkeypatch.setenv("KATOME_INGEST_THREADS", threads)
        with pytest.raises(KatomePanic) as e:
            ingest_files([str(path)], ft, k)
        assert e.value.name == "E_SHORT_READ", threads


def _mutated_text(rng, fastq):
    """a small FASTQ / FASTA text with the irregularities real files have (and a few no file should have)"""
    lines = []
    n_rec = int(rng.integers(0, 14))
    for i in range(n_rec):
        n = int(rng.integers(0, 40)) if rng.random() < 0.15 else int(rng.integers(6, 60))
        s = "".join("ACGT"[c] for c in rng.integers(0, 4, n))
        r = rng.random()
        if r < 0.10 and n:
            p = int(rng.integers(0, n)); s = s[:p] + "N" + s[p + 1:]
        elif r < 0.15 and n:
            s = s.lower()
        elif r < 0.20:
            s = s + "  "
        if fastq:
            q = "".join(chr(int(c)) for c in rng.integers(33, 74, len(s)))
            if rng.random() < 0.2 and q:
                q = "@" + q[1:]
            rec = ["@r%d extra words" % i, s, "+" if rng.random() < 0.8 else "+r%d" % i, q]
        else:
            w = int(rng.integers(5, 30))
            rec = [">r%d" % i] + ([s[j:j + w] for j in range(0, len(s), w)] or ([""] if rng.random() < 0.5 else []))
        lines += rec
    # damage
    for _ in range(int(rng.integers(0, 3))):
        if not lines:
            break
        kind = rng.integers(0, 6)
        p = int(rng.integers(0, len(lines)))
        if kind == 0:
            del lines[p]                                   # a line goes missing
        elif kind == 1:
            lines.insert(p, "")                            # a blank line
        elif kind == 2:
            lines[p] = "x" + lines[p][1:] if lines[p] else "x"   # a first character that is not @ + >
        elif kind == 3:
            lines = lines[:p]                              # the file ends early
        elif kind == 4:
            lines[p] = lines[p] + "\r"                     # one CRLF line
        else:
            lines.insert(p, lines[p])                      # a line twice
    sep = "\r\n" if rng.random() < 0.1 else "\n"
    text = sep.join(lines)
    if lines and rng.random() < 0.8:
        text += sep
    return text


@pytest.mark.parametrize("seed", range(30))
def test_ingest_fuzz_against_oracle(tmp_path, monkeypatch, oracle, seed):
    """irregular and damaged FASTQ / FASTA text: the product's scan -- on one thread and cut into many pieces -- and the
    oracle's parser (both restate bio 0.10's readers) accept the same reads or stop with the same status"""
    rng = np.random.default_rng(9000 + seed)
    k = 5
    for case in range(40):
        fastq = bool(rng.integers(0, 2))
        ft = InputFileType.Fastq if fastq else InputFileType.Fasta
        path = tmp_path / ("f%d.%s" % (case, "fq" if fastq else "fa"))
        path.write_bytes(_mutated_text(rng, fastq).encode())
        try:
            want = ("ok", oracle.build_files([str(path)], k, False, file_type=int(ft)).read_bytes)
            reads = oracle.scan_files([str(path)], file_type=int(ft))
        except oracle.OracleError as e:
            want, reads = ("error", e.name), None
        for threads, chunk in (("1", None), ("4", "8"), ("7", "1")):
            monkeypatch.setenv("KATOME_INGEST_THREADS", threads)
            if chunk:
                monkeypatch.setenv("KATOME_INGEST_MIN_CHUNK", chunk)
            else:
                monkeypatch.delenv("KATOME_INGEST_MIN_CHUNK", raising=False)
            try:
                r = ingest_files([str(path)], ft, k)
                got = ("ok", r["read_bytes"])
            except KatomePanic as e:
                r, got = None, ("error", e.name)
            assert got == want, (seed, case, threads, path.read_bytes())
            if r is not None:
                assert (r["n_records"], r["n_reads"]) == (reads["n_records"], reads["n_accepted"])
                oracle.set_k(k)
                for i in range(r["n_reads"]):
                    seq = bytes(reads["seq"][reads["off"][i]:reads["off"][i + 1]])
                    assert bytes(r["packed"][r["byte_off"][i]:r["byte_off"][i + 1]]) == oracle.compress_node(seq) and r["len"][i] == len(seq)
