"""Shared test helpers (pure Python): k-mer <-> integer conversions and the host build of kmer_bits.h."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def kmer_to_int(s):
    v = 0
    for ch in s:
        v = (v << 2) | CODE[ch]
    return v


def int_to_kmer(v, k):
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp_str(s):
    return s.encode().translate(COMP)[::-1].decode()


def words_to_int(words):
    v = 0
    for w in words:
        v = (v << 64) | int(w)
    return v


def int_to_words(v, nw):
    return [(v >> (64 * (nw - 1 - i))) & 0xFFFFFFFFFFFFFFFF for i in range(nw)]


def pack_reads_ascii(reads):
    """ASCII rows [n, L] (uint8, ACGT only) -> packed bytes [n, ceil(L/4)] in compress_node bit order"""
    reads = np.asarray(reads, dtype=np.uint8)
    n, L = reads.shape
    lut = np.zeros(256, np.uint8)
    for ch, c in CODE.items():
        lut[ord(ch)] = c
    codes = lut[reads]
    pad = (-L) % 4
    if pad:
        codes = np.concatenate([codes, np.zeros((n, pad), np.uint8)], axis=1)
    c4 = codes.reshape(n, -1, 4)
    return (c4[:, :, 0] << 6 | c4[:, :, 1] << 4 | c4[:, :, 2] << 2 | c4[:, :, 3]).astype(np.uint8)


def windows_multiset(reads_ascii, k, rc):
    """reference semantics on strings: every window of every accepted read (+ of its reverse complement)"""
    out = {}
    for row in reads_ascii:
        s = bytes(row).decode()
        if any(ch not in "ACGT" for ch in s):
            continue
        for strand in ((s, revcomp_str(s)) if rc else (s,)):
            for w in range(len(strand) - k + 1):
                km = strand[w:w + k]
                out[km] = out.get(km, 0) + 1
    return out


_SHIM = None


def hostshim():
    global _SHIM
    if _SHIM is None:
        src = os.path.join(HERE, "hostshim", "kmer_bits_host.cpp")
        hdr = os.path.join(ROOT, "katome_amd", "csrc", "kmer_bits.h")
        so = os.path.join(HERE, "hostshim", "libkmer_bits_host.so")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
        L = C.CDLL(so)
        u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
        L.hs_extract.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p]
        L.hs_extract_aligned.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, u64p]
        for f in ("hs_revcomp", "hs_canonical"):
            getattr(L, f).argtypes = [u64p, C.c_uint32, u64p]
        L.hs_endpoints.argtypes = [u64p, C.c_uint32, u64p, u64p]
        L.hs_label.argtypes = [u64p, C.c_uint32, u8p]
        L.hs_hash.argtypes = [u64p, C.c_int]
        L.hs_hash.restype = C.c_uint64
        L.hs_owner.argtypes = [u64p, C.c_int, C.c_uint64]
        L.hs_owner.restype = C.c_uint64
        L.hs_core_owner.argtypes = [u64p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64]
        L.hs_core_owner.restype = C.c_uint64
        L.hs_digit.argtypes = [u64p, C.c_int, C.c_uint32, C.c_uint32]
        L.hs_digit.restype = C.c_uint32
        L.hs_splitmix64.argtypes = [C.c_uint64]
        L.hs_splitmix64.restype = C.c_uint64
        for f in ("hs_mix64", "hs_unmix64"):
            getattr(L, f).argtypes = [C.c_uint64]
            getattr(L, f).restype = C.c_uint64
        _SHIM = L
    return _SHIM


def shim_words(v, nw):
    return (C.c_uint64 * nw)(*int_to_words(v, nw))
