"""Random small builds against the oracle: k, read length, strand mode, numbering, batch sizes drawn at random, so that
tile plans, key-word boundaries (k = 31/32/33, tiles of 63/64/95 bases) and table growth are met in combinations the
fixed test parameters do not list.  usage: python tools/fuzz_build.py [cases=150] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from helpers import pack_reads_ascii, kmer_to_int
from oracle import oracle as o
from katome_amd import device as kd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    k = int(rng.choice([3, 4, 5, 8, 15, 16, 17, 30, 31, 32, 33, 34, 40, 47, 48, 55, 62, 63])) if rng.random() < 0.7 else int(rng.integers(3, 64))
    L = k + int(rng.choice([0, 1, 2, 3, 13, 29, 30, 31, 32, 33, 59, 64, 87, 119, 120])) if rng.random() < 0.7 else k + int(rng.integers(0, 140))
    n = int(rng.integers(1, 2500))
    glen = max(L + 1, int(rng.choice([L + 5, 300, 3000, 40000])))
    rc, first_seen = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    prune = first_seen and rng.random() < 0.5
    stages = str(rng.choice(["d", "w", "dw", "wd", "dc", "cd", "wcd", "dcwc", "dcwce", "dcwced"])) if (first_seen and L > k and rng.random() < 0.6) else None
    thr = int(rng.choice([1, 2, 3, 5]))
    npct = int(rng.choice([0, 0, 3]))
    reads = o.synth_reads(int(rng.integers(0, 1000)), n, L, glen, float(rng.choice([0.0, 1e-3, 2e-2])), npct)
    has_n = (reads == ord("N")).any(axis=1)
    clean = reads.copy(); clean[clean == ord("N")] = ord("C")
    packed = torch.from_numpy(pack_reads_ascii(clean).reshape(-1).copy()).cuda()
    skip = torch.from_numpy(has_n.astype(np.uint8)).cuda()
    b = kd.Builder(k, rc, first_seen_order=first_seen, table_slots_hint=int(rng.choice([0, 1 << 10, 1 << 16])))
    step = int(rng.choice([n, max(1, n // 3), 64]))
    try:
        for r0 in range(0, n, step):
            b.count_reads(packed, min(step, n - r0), L, skip, first_read=r0)
        dg = b.finalize()
        if stages:                                        # the stages of assemble_with_graph, in a random order
            o.set_genome_length(glen)
            for st in stages:
                if st == "d":
                    b.remove_dead_paths()
                elif st == "c":
                    b.standardize_contigs()
                elif st == "w":
                    b.remove_weak_edges(thr)
                else:
                    b.standardize_edges(glen, thr)
            dg = b.graph()
            ref = o.build_ascii(reads, k, rc, remove_weak_edges=thr, stages=stages)
        else:
            if prune:
                dg, _ = b.remove_dead_paths()
            ref = o.build_ascii(reads, k, rc, remove_dead_paths=prune)
        ok = (dg.n_nodes, dg.n_edges) == (ref.n_nodes, ref.n_edges)
        if ok and first_seen:
            ok = (np.array_equal(dg.edge_label.cpu().numpy(), ref.edge_label) and
                  np.array_equal(dg.edge_weight.cpu().numpy().view(np.uint32), ref.edge_weight) and
                  np.array_equal(dg.edge_src.cpu().numpy().view(np.uint64), ref.edge_src) and
                  np.array_equal(dg.edge_dst.cpu().numpy().view(np.uint64), ref.edge_dst))
        elif ok:
            nw = dg.edge_key.shape[1]
            ek = dg.edge_key.cpu().numpy().view(np.uint64)
            keys = [int(r[0]) if nw == 1 else (int(r[0]) << 64) | int(r[1]) for r in ek]
            ok = dict(zip(keys, dg.edge_weight.cpu().numpy().view(np.uint32).tolist())) == {kmer_to_int(s): w for s, w in ref.multiset()}
    finally:
        b.close()
    if not ok:
        bad += 1
        print("MISMATCH case %d: k=%d L=%d n=%d glen=%d rc=%s first_seen=%s prune=%s stages=%s thr=%d step=%d" % (c, k, L, n, glen, rc, first_seen, prune, stages, thr, step), flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
