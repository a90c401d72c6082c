"""katome_amd/csrc/mem_pool.h (the device allocator's bookkeeping) over a fake backend on the host: blocks never overlap,
freed neighbours merge back into whole segments, later buffers are cut out of earlier segments instead of new ones."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MiB, GiB = 1 << 20, 1 << 30


_COPIES = [0]


def _shim():
    _COPIES[0] += 1
    src = os.path.join(HERE, "hostshim", "mem_pool_host.cpp")
    hdr = os.path.join(ROOT, "katome_amd", "csrc", "mem_pool.h")
    so = os.path.join(HERE, "hostshim", "libmem_pool_host_%d_%d.so" % (os.getpid(), _COPIES[0]))      # one pool per library image: a fresh copy per test
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    lib = C.CDLL(so)
    os.unlink(so)
    lib.hs_pool_alloc.restype = C.c_uint64
    lib.hs_pool_alloc.argtypes = [C.c_uint64, C.c_int]
    lib.hs_pool_free.argtypes = [C.c_uint64, C.c_int]
    lib.hs_pool_round.restype = C.c_uint64
    lib.hs_pool_round.argtypes = [C.c_uint64]
    lib.hs_pool_budget.argtypes = [C.c_uint64]
    return lib


def stats(lib):
    out = (C.c_uint64 * 7)()
    lib.hs_pool_stats(out)
    return dict(zip(("backend", "allocs", "free", "segments", "live", "free_blocks", "syncs"), (int(x) for x in out)))


@pytest.fixture()
def lib():
    return _shim()


def test_later_buffers_are_cut_from_earlier_segments(lib):
    table, seen = lib.hs_pool_alloc(36 * GiB, 0), lib.hs_pool_alloc(36 * GiB, 0)
    assert stats(lib)["allocs"] == 2
    lib.hs_pool_free(table, 0); lib.hs_pool_free(seen, 0)
    a = [lib.hs_pool_alloc(12 * GiB, 0) for _ in range(4)] + [lib.hs_pool_alloc(6 * GiB, 0)]
    s = stats(lib)
    assert s["allocs"] == 2 and s["backend"] == 72 * GiB                 # nothing new from the backend
    spans = sorted((p, p + lib.hs_pool_round(n)) for p, n in zip(a, [12 * GiB] * 4 + [6 * GiB]))
    assert all(x[1] <= y[0] for x, y in zip(spans, spans[1:]))           # disjoint
    for p in a:
        assert lib.hs_pool_free(p, 0) == 1
    s = stats(lib)
    assert s["free_blocks"] == 2 and s["free"] == 72 * GiB and s["live"] == 0     # merged back into the two segments
    big = lib.hs_pool_alloc(36 * GiB, 0)
    assert big in (table, seen) and stats(lib)["allocs"] == 2
    lib.hs_pool_free(big, 0)
    lib.hs_pool_release()
    s = stats(lib)
    assert s["backend"] == 0 and s["free"] == 0 and s["segments"] == 0


def test_small_requests_do_not_pin_large_segments(lib):
    big = lib.hs_pool_alloc(4 * GiB, 0)
    lib.hs_pool_free(big, 0)
    small = lib.hs_pool_alloc(64, 0)                                      # its own segment, not the front of the big one
    assert stats(lib)["allocs"] == 2
    assert lib.hs_pool_alloc(4 * GiB, 0) == big and stats(lib)["allocs"] == 2
    lib.hs_pool_free(small, 0)
    assert lib.hs_pool_alloc(100, 0) == small                             # exact rounded size comes back
    assert lib.hs_pool_alloc(5000, 0) != small and stats(lib)["allocs"] == 3


def test_a_remainder_not_worth_keeping_stays_with_the_block(lib):
    p = lib.hs_pool_alloc(1 * GiB, 0)
    lib.hs_pool_free(p, 0)
    q = lib.hs_pool_alloc(1 * GiB - 16 * MiB, 0)
    assert q == p and stats(lib)["free_blocks"] == 0


def test_out_of_memory_gives_free_segments_back_first(lib):
    lib.hs_pool_budget(10 * GiB)
    a, b = lib.hs_pool_alloc(4 * GiB, 0), lib.hs_pool_alloc(4 * GiB, 0)
    lib.hs_pool_free(a, 0)
    c = lib.hs_pool_alloc(5 * GiB, 0)                                     # does not fit beside the cached 4 GiB: that one is released
    assert c != 0 and stats(lib)["backend"] == 9 * GiB
    assert lib.hs_pool_alloc(5 * GiB, 0) == 0                             # truly out
    lib.hs_pool_free(b, 0); lib.hs_pool_free(c, 0)
    assert lib.hs_pool_free(12345, 0) == 0                                # not ours


def test_streams(lib):
    p = lib.hs_pool_alloc(1 * GiB, 1)
    lib.hs_pool_free(p, 1)
    assert lib.hs_pool_alloc(1 * GiB, 1) == p and stats(lib)["syncs"] == 0
    lib.hs_pool_free(p, 1)
    assert lib.hs_pool_alloc(1 * GiB, 2) == p and stats(lib)["syncs"] == 1   # another stream waits for the last user


@pytest.mark.parametrize("seed", range(6))
def test_random_traffic(lib, seed):
    rng = np.random.default_rng(seed)
    live = {}
    for step in range(4000):
        if live and (rng.random() < 0.48 or len(live) > 60):
            p = list(live)[int(rng.integers(0, len(live)))]
            assert lib.hs_pool_free(p, int(rng.integers(0, 2))) == 1
            del live[p]
        else:
            n = int(rng.choice([64, 4096, 3 * MiB, 9 * MiB, 40 * MiB, 300 * MiB, 2 * GiB, 7 * GiB]) * (0.5 + rng.random()))
            p = lib.hs_pool_alloc(n, int(rng.integers(0, 2)))
            assert p and p not in live
            live[p] = lib.hs_pool_round(n)
        if step % 97 == 0:
            spans = sorted((p, p + n) for p, n in live.items())
            assert all(x[1] <= y[0] for x, y in zip(spans, spans[1:]))
            s = stats(lib)
            assert s["live"] == len(live) and s["segments"] == s["backend"]
            assert s["free"] + sum(live.values()) <= s["segments"]          # (<=: remainders not worth keeping ride along)
    for p in list(live):
        lib.hs_pool_free(p, 0)
    s = stats(lib)
    assert s["live"] == 0 and s["free"] == s["segments"] == s["backend"]    # everything merged back
    lib.hs_pool_release()
    assert stats(lib)["backend"] == 0
