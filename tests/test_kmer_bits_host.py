"""The kernels' bit arithmetic (katome_amd/csrc/kmer_bits.h, compiled for the host) against the
oracle's restatement of compress.rs -- runs without a GPU."""
import ctypes as C
import random

import pytest

from helpers import hostshim, int_to_kmer, kmer_to_int, revcomp_str, shim_words, words_to_int

KS = [3, 4, 5, 7, 16, 30, 31, 32, 33, 40, 47, 62, 63]


def nw_of(k):
    return 1 if 2 * k <= 62 else 2


@pytest.mark.parametrize("k", KS)
def test_window_extraction_matches_compress_kmer(oracle, k):
    """key of window w == the ASCII k-mer; its halves == the two nodes of compress_kmer (compress.rs:18-28)"""
    L = hostshim()
    rng = random.Random(k)
    oracle.set_k(k)
    nw = nw_of(k)
    for _ in range(20):
        n = rng.randrange(k, k + 70)
        read = "".join(rng.choice("ACGT") for _ in range(n))
        packed = oracle.compress_node(read.encode())          # compress_node bit order == packed read
        for w in range(0, n - k + 1):
            out = (C.c_uint64 * nw)()
            L.hs_extract(packed, len(packed), 0, w, k, out)
            key = words_to_int(out)
            assert key == kmer_to_int(read[w:w + k]), (k, w)
            out2 = (C.c_uint64 * nw)()
            L.hs_extract_aligned(packed, len(packed), 2 * w, k, out2)
            assert words_to_int(out2) == key
            src, dst = (C.c_uint64 * nw)(), (C.c_uint64 * nw)()
            L.hs_endpoints(out, k, src, dst)
            ck = oracle.compress_kmer(read[w:w + k].encode())
            half = len(ck) // 2
            pad = 2 * ((4 - (k - 1) % 4) % 4)
            assert words_to_int(src) << pad == int.from_bytes(ck[:half], "big")
            assert words_to_int(dst) << pad == int.from_bytes(ck[half:], "big")


@pytest.mark.parametrize("k", KS)
def test_revcomp_matches_compress_kmer_with_rev_compl(oracle, k):
    L = hostshim()
    rng = random.Random(100 + k)
    oracle.set_k(k)
    nw = nw_of(k)
    for _ in range(200):
        s = "".join(rng.choice("ACGT") for _ in range(k))
        out = (C.c_uint64 * nw)()
        L.hs_revcomp(shim_words(kmer_to_int(s), nw), k, out)
        rc = words_to_int(out)
        assert int_to_kmer(rc, k) == revcomp_str(s)
        # oracle: rc k-mer in compress_kmer format decompresses to the same string
        _, rev = oracle.compress_kmer_with_rev_compl(s.encode())
        assert oracle.decompress_kmer(rev).decode() == int_to_kmer(rc, k)
        can = (C.c_uint64 * nw)()
        L.hs_canonical(shim_words(kmer_to_int(s), nw), k, can)
        assert words_to_int(can) == min(kmer_to_int(s), rc)


def test_palindromes_only_for_even_k():
    L = hostshim()
    for k in (4, 6, 32):
        s = "ACGT" * (k // 4)
        s = s[:k // 2]
        pal = s + revcomp_str(s)
        nw = nw_of(k)
        out = (C.c_uint64 * nw)()
        L.hs_revcomp(shim_words(kmer_to_int(pal), nw), k, out)
        assert words_to_int(out) == kmer_to_int(pal)


@pytest.mark.parametrize("k", KS)
def test_label_matches_compress_edge(oracle, k):
    """label == compress_edge(k-mer) == kmer_to_edge(compress_kmer(k-mer)) (pt_graph.rs:339-343)"""
    L = hostshim()
    rng = random.Random(200 + k)
    oracle.set_k(k)
    nw = nw_of(k)
    stride = 1 + (k + 3) // 4
    for _ in range(100):
        s = "".join(rng.choice("ACGT") for _ in range(k))
        buf = (C.c_uint8 * stride)()
        L.hs_label(shim_words(kmer_to_int(s), nw), k, buf)
        assert bytes(buf) == oracle.compress_edge(s.encode())
        assert bytes(buf) == oracle.kmer_to_edge(oracle.compress_kmer(s.encode()))


def test_digits_and_owner_ranges():
    L = hostshim()
    rng = random.Random(5)
    for nw in (1, 2):
        for _ in range(200):
            v = rng.getrandbits(62 if nw == 1 else 126)
            w = shim_words(v, nw)
            for shift in range(0, 64 * nw - 8, 8):
                assert L.hs_digit(w, nw, shift, 8) == (v >> shift) & 0xFF
            assert L.hs_digit(w, nw, 3, 11) == (v >> 3) & 0x7FF
            for n in (1, 2, 3, 8):
                assert 0 <= L.hs_owner(w, nw, n) < n


def test_owner_of_a_record_is_independent_of_its_table_slot():
    """a rank's table puts a key at mulhi(hash, capacity); if the owner were mulhi(hash, ranks) as well, every key of a rank would
    land in one n-th of its table (eight thread ranks: a mid-tile insert of 1.6 M records took 2 s).  The keys one rank owns
    spread evenly over the table's stretches."""
    L = hostshim()
    rng = random.Random(11)
    ranks, stretches = 8, 16
    for nw in (1, 2, 3):
        hist = [[0] * stretches for _ in range(ranks)]
        for _ in range(16000):
            w = shim_words(rng.getrandbits(60 * nw), nw)
            hist[L.hs_owner(w, nw, ranks)][(L.hs_hash(w, nw) * stretches) >> 64] += 1
        for r in range(ranks):
            n = sum(hist[r])
            assert 1600 < n < 2400
            assert max(hist[r]) < 2.0 * n / stretches and min(hist[r]) > 0.5 * n / stretches, (nw, r, hist[r])


def test_the_group_hash_of_one_word_keys_is_a_bijection():
    """lds_count_packed_kernel (table.hip) keeps a one-word k-mer as the low 48 bits of mix64(k-mer) inside the hash group the top 16
    bits name, and reads it back with unmix64: the pair must be inverse to each other on all 64-bit words, and one-word keys of a
    group must differ in their low 48 bits"""
    L = hostshim()
    rng = random.Random(64)
    words = [0, 1, (1 << 62) - 1, (1 << 64) - 1] + [rng.getrandbits(64) for _ in range(20000)] + [rng.getrandbits(62) for _ in range(20000)]
    for x in words:
        assert L.hs_unmix64(L.hs_mix64(x)) == x and L.hs_mix64(L.hs_unmix64(x)) == x
    seen = {}
    for x in words:                      # (group, remainder) names the key
        h = L.hs_mix64(x)
        assert seen.setdefault((h >> 48, h & ((1 << 48) - 1)), x) == x
    a = (C.c_uint64 * 1)(words[7])
    assert L.hs_hash(a, 1) == L.hs_mix64(words[7])           # (the partition passes and the group index hash one-word keys with it)


def test_splitmix_matches_oracle(oracle):
    L = hostshim()
    for x in (0, 1, 2 ** 63, 0x6B61746F6D650001, 2 ** 64 - 1):
        assert L.hs_splitmix64(x) == oracle.splitmix64(x)


@pytest.mark.parametrize("k", [3, 4, 5, 16, 31, 32, 33, 40, 63])
def test_core_owner_keeps_a_node_with_its_out_edges(k):
    """kmer_bits.h core_owner: a k-mer, its reverse complement and the source nodes of both land on the same rank
    (what lets the multi-GPU build read source ids off each rank's own edges); owners spread over the ranks"""
    import random
    from helpers import revcomp_str
    L = hostshim()
    rng = random.Random(k)
    nw = 1 if 2 * k <= 62 else 2
    seen = set()
    for _ in range(300):
        s = "".join(rng.choice("ACGT") for _ in range(k))
        x, r = kmer_to_int(s), kmer_to_int(revcomp_str(s))
        for n in (2, 3, 8):
            o = L.hs_core_owner(shim_words(x, nw), nw, 2, k - 2, n)
            assert 0 <= o < n
            assert o == L.hs_core_owner(shim_words(r, nw), nw, 2, k - 2, n)            # the stored canonical k-mer decides for both
            assert o == L.hs_core_owner(shim_words(x >> 2, nw), nw, 0, k - 2, n)       # source node of x
            assert o == L.hs_core_owner(shim_words(r >> 2, nw), nw, 0, k - 2, n)       # source node of rc(x)
            if n == 8:
                seen.add(o)
        # all four out-edges of a node share the owner
        node = x >> 2
        owners = {L.hs_core_owner(shim_words((node << 2) | b, nw), nw, 2, k - 2, 8) for b in range(4)}
        assert len(owners) == 1
    assert len(seen) >= (2 if k <= 4 else 6)


@pytest.mark.parametrize("kk", [64, 65, 73, 84, 94, 95])
def test_three_word_tiles(kk):
    """tiles of k-mers may take three words (64..95 bases): extraction, reverse complement, canonical form and the
    sub-windows the expansion kernels take out of them, against plain string arithmetic"""
    from helpers import pack_reads_ascii
    import numpy as np
    L = hostshim()
    L.hs_sub_window.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    assert L.hs_key_words(kk) == 3 and L.hs_key_words(63) == 2 and L.hs_key_words(31) == 1
    rng = random.Random(kk)
    for _ in range(12):
        n = rng.randrange(kk, kk + 60)
        read = "".join(rng.choice("ACGT") for _ in range(n))
        packed = pack_reads_ascii(np.frombuffer(read.encode(), np.uint8).reshape(1, -1)).reshape(-1).tobytes()
        for w in range(0, n - kk + 1, 3):
            out = (C.c_uint64 * 3)()
            L.hs_extract(packed, len(packed), 0, w, kk, out)
            tile = words_to_int(out)
            assert tile == kmer_to_int(read[w:w + kk]), (kk, w)
            out2 = (C.c_uint64 * 3)()
            L.hs_extract_aligned(packed, len(packed), 2 * w, kk, out2)
            assert words_to_int(out2) == tile
            rc = (C.c_uint64 * 3)()
            L.hs_revcomp(out, kk, rc)
            assert int_to_kmer(words_to_int(rc), kk) == revcomp_str(read[w:w + kk])
            can = (C.c_uint64 * 3)()
            L.hs_canonical(out, kk, can)
            assert words_to_int(can) == min(tile, words_to_int(rc))
    # sub-windows: a tile of `span` k-mers -> its k-mers (two words or one), and a big tile -> its mid tiles (three words)
    for k, span in ((63, kk - 62), (40, kk - 39), (kk - 32, 33)):
        if span < 2 or k + span - 1 != kk or k > 63:
            continue
        s = "".join(rng.choice("ACGT") for _ in range(kk))
        tile = shim_words(kmer_to_int(s), 3)
        nwk = nw_of(k)
        for o in range(span):
            out = (C.c_uint64 * nwk)()
            L.hs_sub_window(tile, 3, nwk, k, span, 1, o, out)
            assert int_to_kmer(words_to_int(out), k) == s[o:o + k]
        for s2 in range(2, span):
            if span % s2 == 0 and k + s2 - 1 >= 64:           # mid tiles that still need three words
                for o in range(span // s2):
                    out = (C.c_uint64 * 3)()
                    L.hs_sub_window(tile, 3, 3, k + s2 - 1, span // s2, s2, o, out)
                    assert int_to_kmer(words_to_int(out), k + s2 - 1) == s[o * s2:o * s2 + k + s2 - 1]
