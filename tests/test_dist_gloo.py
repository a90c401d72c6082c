"""world_size-2 (and 3) runs of the multi-GPU protocol (katome_amd/dist.py) on CPU with gloo.
The device primitives are replaced by a numpy test double (tests/numpy_ops.py); what is under test is
the host logic: read sharding, routing by owner, variable-size all-to-all, global node numbering.
The merged result of all ranks must equal the oracle's single-process build, whatever the world size."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, rc, n_reads, read_len, batch_reads, out_dir, min_weight=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from helpers import pack_reads_ascii, words_to_int
    from numpy_ops import NumpyOps
    from katome_amd import dist as kdist
    from oracle import oracle as o
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ascii_reads = o.synth_reads(0, n_reads, read_len, 3000, 2e-2, 4)
        has_n = (ascii_reads == ord("N")).any(axis=1)
        clean = ascii_reads.copy()
        clean[clean == ord("N")] = ord("A")
        r0, r1 = kdist.shard_range(n_reads, world, rank)
        # every rank packs only its own shard
        packed = torch.from_numpy(pack_reads_ascii(clean[r0:r1]).reshape(-1).copy()) if r1 > r0 else torch.zeros(0, dtype=torch.uint8)
        skip = torch.from_numpy(has_n[r0:r1].astype(np.uint8))
        ops = NumpyOps(k, rc, min_weight)
        kdist.build_shard(ops, packed, skip, r1 - r0, read_len, batch_reads)
        g = kdist.finalize_distributed(ops)
        nw = ops.nw
        ek = g.edge_key.numpy().view(np.uint64).reshape(-1, nw)
        nk = g.node_key.numpy().view(np.uint64).reshape(-1, nw)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                 edge_key=np.array([words_to_int(r) for r in ek], dtype=object),
                 weight=g.edge_weight.numpy(), src=g.edge_src.numpy(), dst=g.edge_dst.numpy(),
                 label=g.edge_label.numpy(), node_key=np.array([words_to_int(r) for r in nk], dtype=object),
                 node_base=g.node_base, total_nodes=g.total_nodes, total_edges=g.total_edges)
    finally:
        dist.destroy_process_group()


# katome_tile_plan: read_len 50: k=11 -> 40 windows = 2 tiles of 20; k=33 -> 18 windows = 1 tile of 18 (128-bit);
# k=12 -> 39 windows = 3 tiles of 13; k=6 -> 45 windows = 3 tiles of 15; read_len 53 with k=11: 43 windows (prime) =
# 2 tiles of 21 + 1 window left over (the left-over windows travel as plain k-mer records); k=60: no span fits 63 bases
@pytest.mark.parametrize("world,k,rc,n_reads,batch,read_len", [(2, 11, True, 260, 64, 50), (2, 33, True, 130, 1000, 50),
                                                                (3, 12, False, 200, 64, 50), (2, 6, True, 70, 64, 50),
                                                                (2, 11, True, 130, 64, 53), (3, 11, False, 150, 40, 57),
                                                                (2, 60, True, 60, 64, 75),
                                                                (3, 11, True, 100, 40, 53),      # the third rank gets no reads at all
                                                                (3, 12, False, 60, 64, 50),      # two ranks without reads
                                                                (8, 11, True, 700, 64, 50),      # the node's eight ranks
                                                                (8, 33, False, 200, 1000, 50)])  # ... some of them idle
def test_distributed_build_equals_oracle(oracle, tmp_path, world, k, rc, n_reads, batch, read_len):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, rc, n_reads, read_len, batch, str(tmp_path)), nprocs=world, join=True)
    ref = oracle.build_ascii(oracle.synth_reads(0, n_reads, read_len, 3000, 2e-2, 4), k, rc)
    from helpers import int_to_kmer
    merged, node_of_id = {}, {}
    total_nodes = total_edges = None
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r), allow_pickle=True) for r in range(world)]
    for p in parts:
        total_nodes, total_edges = int(p["total_nodes"]), int(p["total_edges"])
        for i, key in enumerate(p["node_key"]):
            gid = int(p["node_base"]) + i
            assert gid not in node_of_id
            node_of_id[gid] = int(key)
    mask = (1 << (2 * (k - 1))) - 1
    for p in parts:
        for key, w, s, d, lab in zip(p["edge_key"], p["weight"], p["src"], p["dst"], p["label"]):
            key = int(key)
            assert key not in merged                           # every k-mer lives on exactly one rank
            merged[key] = int(np.uint32(w))
            assert node_of_id[int(s)] == key >> 2 and node_of_id[int(d)] == key & mask
            oracle.set_k(k)
            assert bytes(lab) == oracle.compress_edge(int_to_kmer(key, k).encode())
    assert sorted((int_to_kmer(v, k), w) for v, w in merged.items()) == ref.multiset()
    assert (total_nodes, total_edges) == (ref.n_nodes, ref.n_edges)
    assert sorted(node_of_id) == list(range(ref.n_nodes))       # dense global numbering
    assert len(set(node_of_id.values())) == ref.n_nodes


@pytest.mark.parametrize("world,k,rc,threshold", [(2, 11, True, 2), (3, 12, False, 3)])
def test_distributed_build_with_weak_edges_removed(oracle, tmp_path, world, k, rc, threshold):
    """Clean::remove_weak_edges (pruner.rs:84-93) on the sharded build: the weights are complete at the k-mers' owners, the
    edges under the threshold are never read out and the nodes they leave alone never get an id -- same edge multiset and
    node count as the oracle's build + remove_weak_edges"""
    n_reads, read_len = 220, 50
    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, rc, n_reads, read_len, 64, str(tmp_path), threshold), nprocs=world, join=True)
    ref = oracle.build_ascii(oracle.synth_reads(0, n_reads, read_len, 3000, 2e-2, 4), k, rc, remove_weak_edges=threshold)
    from helpers import int_to_kmer
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r), allow_pickle=True) for r in range(world)]
    merged, nodes = {}, set()
    mask = (1 << (2 * (k - 1))) - 1
    for p in parts:
        node_of = {int(p["node_base"]) + i: int(key) for i, key in enumerate(p["node_key"])}
        nodes.update(node_of.values())
        for key, w in zip(p["edge_key"], p["weight"]):
            assert int(key) not in merged and int(np.uint32(w)) >= threshold
            merged[int(key)] = int(np.uint32(w))
        assert (int(p["total_nodes"]), int(p["total_edges"])) == (ref.n_nodes, ref.n_edges)
    assert sorted((int_to_kmer(v, k), w) for v, w in merged.items()) == ref.multiset()
    assert nodes == {v >> 2 for v in merged} | {v & mask for v in merged} and len(nodes) == ref.n_nodes
    assert 0 < ref.n_edges < oracle.build_ascii(oracle.synth_reads(0, n_reads, read_len, 3000, 2e-2, 4), k, rc).n_edges


def _exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from katome_amd import dist as kdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        kdist.MAX_MESSAGE_BYTES = 64                      # force the multi-round path: 4 records of 2 words per round
        nw = 2
        counts = [(7 * rank + 5 * p + 3) % 23 for p in range(world)]          # records for each destination
        send = torch.cat([torch.arange(c * nw, dtype=torch.int64) + 1000 * p + 100000 * rank for p, c in enumerate(counts)])
        recv, rcounts = kdist._exchange(send, counts, nw, None)
        want_counts = [(7 * src + 5 * rank + 3) % 23 for src in range(world)]
        assert rcounts == want_counts
        want = torch.cat([torch.arange(c * nw, dtype=torch.int64) + 1000 * rank + 100000 * src for src, c in enumerate(want_counts)])
        assert torch.equal(recv, want)
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_in_rounds(tmp_path, world):
    """messages above the per-pair limit are cut into rounds and reassembled in place"""
    mp.spawn(_exchange_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def test_shard_range_covers_all_reads():
    from katome_amd.dist import shard_range
    for total in (0, 1, 63, 64, 65, 1000, 200_000_000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(s[0] % 64 == 0 or s[0] == total for s in spans)
