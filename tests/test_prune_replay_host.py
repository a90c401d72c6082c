"""katome_amd/csrc/prune_replay.h (the host half of remove_dead_paths) against a literal Vec::swap_remove simulation."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NONE = 0xFFFFFFFF


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(HERE, "hostshim", "prune_replay_host.cpp")
    hdr = os.path.join(ROOT, "katome_amd", "csrc", "prune_replay.h")
    so = os.path.join(HERE, "hostshim", "libprune_replay_host.so")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run_edges(shim, pos, mult, n_edges):
    pos, mult = np.ascontiguousarray(pos, np.uint32), np.ascontiguousarray(mult, np.uint32)
    marks = int(mult.sum())
    victims, to, frm = np.zeros(marks + 1, np.uint32), np.zeros(len(pos) + 1, np.uint32), np.zeros(len(pos) + 1, np.uint32)
    counts = np.zeros(4, np.uint64)
    shim.hs_replay_edges(_p(pos), _p(mult), C.c_uint64(len(pos)), C.c_uint64(n_edges), C.c_uint64(marks), _p(victims), _p(to), _p(frm),
                         _p(counts))
    nv, nm, n_new, dups = (int(x) for x in counts)
    return victims[:nv].tolist(), dict(zip(to[:nm].tolist(), frm[:nm].tolist())), n_new, dups


def naive_edges(pos, mult, n_edges):
    """remove_paths (pruner.rs:199-217): indices sorted descending, Graph::remove_edge = Vec::swap_remove"""
    arr = list(range(n_edges))
    todo = sorted((int(p) for p, c in zip(pos, mult) for _ in range(int(c))), reverse=True)
    victims = []
    for d in todo:
        if d < len(arr):
            victims.append(arr[d])
            arr[d] = arr[-1]
            arr.pop()
    return victims, {p: v for p, v in enumerate(arr) if v != p}, len(arr)


@pytest.mark.parametrize("seed", range(40))
def test_edge_replay(shim, seed):
    rng = np.random.default_rng(seed)
    n_edges = int(rng.integers(1, 400))
    u = int(rng.integers(0, n_edges + 1))
    pos = np.sort(rng.choice(n_edges, u, replace=False)).astype(np.uint32)
    heavy = seed % 3 == 0                               # many repeated indices, also runs that empty the whole tail
    mult = rng.integers(1, 6 if heavy else 3, u).astype(np.uint32) if u else np.zeros(0, np.uint32)
    if seed % 4:
        mult[rng.random(u) < 0.7] = 1
    victims, moves, n_new, dups = run_edges(shim, pos, mult, n_edges)
    want_v, want_m, want_n = naive_edges(pos, mult, n_edges)
    assert (victims, moves, n_new) == (want_v, want_m, want_n)
    assert dups <= len(victims) and ((mult <= 1).all() <= (dups == 0))
    assert all(t < n_new <= f for t, f in moves.items())      # copies never overlap (prune.hip relies on it)


def test_edge_replay_extremes(shim):
    assert run_edges(shim, [], [], 5) == ([], {}, 5, 0)
    assert run_edges(shim, [4], [3], 5) == ([4], {}, 4, 0)            # the last index twice more: out of range, no-ops
    assert run_edges(shim, [0], [5], 5)[0] == [0, 4, 3, 2, 1]           # index 0 again and again removes whatever moved in
    assert run_edges(shim, list(range(5)), [1] * 5, 5) == ([4, 3, 2, 1, 0], {}, 0, 0)


def run_nodes(shim, die, n_nodes):
    die = np.ascontiguousarray(die, np.uint32).reshape(-1)
    m = len(die) // 2
    to, frm = np.zeros(2 * m + 1, np.uint32), np.zeros(2 * m + 1, np.uint32)
    counts = np.zeros(2, np.uint64)
    shim.hs_replay_nodes(_p(die), C.c_uint64(m), C.c_uint64(n_nodes), _p(to), _p(frm), _p(counts))
    nm, n_new = int(counts[0]), int(counts[1])
    return dict(zip(to[:nm].tolist(), frm[:nm].tolist())), n_new


def naive_nodes(die, n_nodes):
    """remove_single_node per endpoint, larger current index first (pruner.rs:206-215); remove_node = swap_remove"""
    arr = list(range(n_nodes))
    where = {v: v for v in arr}

    def remove(v):
        p = where.pop(v)
        last = arr.pop()
        if p < len(arr):
            arr[p] = last
            where[last] = p

    for a, b in die:
        gone = [v for v in (a, b) if v != NONE]
        gone.sort(key=lambda v: where[v], reverse=True)
        for v in gone:
            remove(v)
    return {p: v for p, v in enumerate(arr) if v != p}, len(arr)


@pytest.mark.parametrize("seed", range(40))
def test_node_replay(shim, seed):
    rng = np.random.default_rng(1000 + seed)
    n_nodes = int(rng.integers(2, 500))
    n_die = int(rng.integers(0, n_nodes + 1))
    dying = rng.permutation(n_nodes)[:n_die].tolist()
    die = []
    while dying:
        kind = rng.integers(0, 4)
        if kind == 0 and len(dying) >= 2:
            die.append((dying.pop(), dying.pop()))
        elif kind == 1:
            die.append((dying.pop(), NONE))
        elif kind == 2:
            die.append((NONE, dying.pop()))
        else:
            die.append((NONE, NONE))
    moves, n_new = run_nodes(shim, die, n_nodes)
    want_m, want_n = naive_nodes(die, n_nodes)
    assert (moves, n_new) == (want_m, want_n)
    assert all(t < n_new <= f for t, f in moves.items())


def test_replays_stream(shim):
    """a large pass finishes at tens of millions of removals per second on one host core (the replays are the serial
    part of remove_dead_paths, so their cost is reported)"""
    rng = np.random.default_rng(7)
    n_edges = 20_000_000
    pos = np.flatnonzero(rng.random(n_edges) < 0.1).astype(np.uint32)
    mult = np.ones(len(pos), np.uint32)
    mult[rng.random(len(pos)) < 0.05] = 2
    t0 = time.perf_counter()
    victims, moves, n_new, dups = run_edges(shim, pos, mult, n_edges)
    dt = time.perf_counter() - t0
    assert n_new == n_edges - len(victims) and dups > 0
    assert dt < 5.0
