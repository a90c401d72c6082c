"""Random small sharded builds (katome_amd/dist.py with the HIP kernels, 2-4 ranks sharing cuda:0, gloo for the
exchanges) against the oracle: k, read length, strands, rank count, batch size, weak-edge threshold drawn at random.
usage: python tools/fuzz_dist.py [cases=20] [seed=0]"""
import os, sys, socket, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, k, rc, n_reads, read_len, batch_reads, glen, err, thr, seed, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from katome_amd import dist as kdist
    from oracle import oracle as o
    from helpers import pack_reads_ascii as pack
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        reads = o.synth_reads(seed, n_reads, read_len, glen, err, 2)
        has_n = (reads == ord("N")).any(axis=1)
        clean = reads.copy(); clean[clean == ord("N")] = ord("A")
        r0, r1 = kdist.shard_range(n_reads, world, rank)
        packed = torch.from_numpy(pack(clean[r0:r1]).reshape(-1).copy()).cuda() if r1 > r0 else torch.zeros(0, dtype=torch.uint8, device="cuda")
        skip = torch.from_numpy(has_n[r0:r1].astype(np.uint8)).cuda()
        ops = kdist.HipOps(k, rc, 0, min_weight=thr)
        kdist.build_shard(ops, packed, skip, r1 - r0, read_len, batch_reads)
        g = kdist.finalize_distributed(ops)
        nw = ops.nw
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
                 edge_key=g.edge_key.cpu().numpy().view(np.uint64).reshape(-1, nw), weight=g.edge_weight.cpu().numpy().view(np.uint32),
                 src=g.edge_src.cpu().numpy(), dst=g.edge_dst.cpu().numpy(),
                 node_key=g.node_key.cpu().numpy().view(np.uint64).reshape(-1, nw), node_base=g.node_base,
                 total_nodes=g.total_nodes, total_edges=g.total_edges)
        ops.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    from oracle import oracle as o
    from helpers import kmer_to_int
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for c in range(cases):
        world = int(rng.integers(2, 5))
        k = int(rng.choice([5, 11, 16, 21, 31, 32, 33, 40, 47, 63]))
        L = k + int(rng.choice([0, 3, 20, 29, 30, 59, 87, 119]))
        n = int(rng.integers(1, 6000))
        rc = bool(rng.integers(0, 2))
        glen = int(rng.choice([L + 10, 3000, 60000]))
        err = float(rng.choice([0.0, 2e-3, 2e-2]))
        thr = int(rng.choice([0, 0, 2, 3]))
        batch = int(rng.choice([64, 512, 4096]))
        seed = int(rng.integers(0, 1000))
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        out = tempfile.mkdtemp()
        mp.spawn(worker, args=(world, port, k, rc, n, L, batch, glen, err, thr, seed, out), nprocs=world, join=True)
        ref = o.build_ascii(o.synth_reads(seed, n, L, glen, err, 2), k, rc, remove_weak_edges=thr if thr else None)
        parts = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]
        to_int = lambda row: int(row[0]) if len(row) == 1 else (int(row[0]) << 64) | int(row[1])
        node_of, merged, ok = {}, {}, True
        mask = (1 << (2 * (k - 1))) - 1
        for p in parts:
            for i, row in enumerate(p["node_key"]):
                node_of[int(p["node_base"]) + i] = to_int(row)
        for p in parts:
            ok &= (int(p["total_nodes"]), int(p["total_edges"])) == (ref.n_nodes, ref.n_edges)
            for j, (row, w) in enumerate(zip(p["edge_key"], p["weight"])):
                key = to_int(row)
                ok &= key not in merged
                merged[key] = int(w)
                if j % 7 == 0:
                    ok &= node_of.get(int(p["src"][j])) == key >> 2 and node_of.get(int(p["dst"][j])) == key & mask
        ok &= sorted(merged.items()) == sorted((kmer_to_int(s), w) for s, w in ref.multiset())
        ok &= sorted(node_of) == list(range(ref.n_nodes)) and len(set(node_of.values())) == ref.n_nodes
        if not ok:
            bad += 1
            print("MISMATCH case %d: world=%d k=%d L=%d n=%d rc=%s glen=%d err=%g thr=%d batch=%d seed=%d" % (c, world, k, L, n, rc, glen, err, thr, batch, seed), flush=True)
    print("%d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)
