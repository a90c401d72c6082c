"""Pin the oracle's codec against every known-answer vector the reference's own tests hold
for src/katome/compress.rs (tests/golden/compress_kat.json cites each one)."""
import json
import os
import random

import pytest


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "compress_kat.json")) as f:
        return json.load(f)


def test_compress_edge(oracle, kat):
    for text, expect in kat["compress_edge"]["cases"]:
        assert list(oracle.compress_edge(text.encode())) == expect


def test_decompress_edge(oracle, kat):
    for data, text in kat["decompress_edge"]["cases"]:
        assert oracle.decompress_edge(bytes(data)) == text.encode()


def test_encode_fasta_symbol(oracle, kat):
    e = kat["encode_fasta_symbol"]
    block = 0
    for sym, code in e["symbols"].items():
        assert oracle.encode_fasta_symbol(ord(sym), 0) == code
    for sym in "ACGT":
        block = oracle.encode_fasta_symbol(ord(sym), block)
    assert block == e["block_ACGT"]


def test_single_chunk_roundtrip(oracle, kat):
    text = kat["single_chunk"]["text"]
    byte = 0
    for c in text.encode():
        byte = oracle.encode_fasta_symbol(c, byte)
    assert oracle.decode_compressed_chunk(byte) == text.encode()


def test_last_char(oracle, kat):
    e = kat["last_char"]
    byte = e["start_carrier"]
    for c in e["text"].encode():
        byte = oracle.encode_fasta_symbol(c, byte)
    assert oracle.decompress_char(byte, e["padding"]) == e["expect"]


def test_add_char_to_edge(oracle, kat):
    for text, ch, expect in kat["add_char_to_edge"]["cases"]:
        assert list(oracle.add_char_to_edge(oracle.compress_edge(text.encode()), ord(ch))) == expect


def test_change_last_char(oracle, kat):
    for data, ch, expect in kat["change_last_char_in_edge"]["cases"]:
        assert list(oracle.change_last_char_in_edge(bytes(data), ord(ch))) == expect


def test_extend_edge(oracle, kat):
    for text, ext, expect in kat["extend_edge"]["cases"]:
        assert list(oracle.extend_edge(oracle.compress_edge(text.encode()), ext.encode())) == expect


def test_shifts(oracle, kat):
    for s, expect in enumerate(kat["shift_right"]["outputs"]):
        assert list(oracle.shift_right_bit_array(kat["shift_right"]["input"], s)) == expect
    for s, expect in enumerate(kat["shift_left"]["outputs"]):
        assert list(oracle.shift_left_bit_array(kat["shift_left"]["input"], s)) == expect


def test_reverse_compressed_node(oracle, kat):
    for data, rem, expect in kat["reverse_compressed_node"]["cases"]:
        got = oracle.reverse_compressed_node(data, rem)
        assert list(got) == expect
        assert list(oracle.reverse_compressed_node(got, rem)) == data   # involution, compress.rs:628


def test_kmer_roundtrip_random(oracle):
    """compress.rs:476-494 (properly_compresses_vertex), over many k."""
    rng = random.Random(7)
    for k in (3, 4, 5, 16, 31, 32, 33, 40, 63, 64):
        oracle.set_k(k)
        for _ in range(50):
            s = bytes(rng.choice(b"ACGT") for _ in range(k))
            assert oracle.decompress_kmer(oracle.compress_kmer(s)) == s
            assert oracle.decompress_edge(oracle.kmer_to_edge(oracle.compress_kmer(s))) == s


def test_rev_compl_is_reverse_complement(oracle):
    """compress_kmer_with_rev_compl (compress.rs:34-48) yields the k-mer of the reverse
    complement: checked against a string-level reverse complement."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rng = random.Random(11)
    for k in (3, 5, 31, 32, 40, 63):
        oracle.set_k(k)
        for _ in range(50):
            s = bytes(rng.choice(b"ACGT") for _ in range(k))
            fwd, rev = oracle.compress_kmer_with_rev_compl(s)
            assert fwd == oracle.compress_kmer(s)
            assert rev == oracle.compress_kmer(s.translate(comp)[::-1])
