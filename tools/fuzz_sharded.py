"""Random sharded builds (thread ranks sharing one card, both routes, both numberings, pruning) against the oracle: world size, k,
read length, strand mode, error rate and read count drawn at random, so that tile plans (one-, two- and three-word tiles, with
and without mid tiles and left-over windows), ranks without reads and both routes meet in combinations the fixed tests do not
list.  usage: python tools/fuzz_sharded.py [cases=120] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import pack_reads_ascii
from oracle import oracle as o
from katome_amd.build import GpuGraph

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    world = int(rng.choice([1, 2, 2, 3, 4, 5, 8]))
    k = int(rng.choice([5, 11, 12, 16, 21, 31, 32, 33, 40, 47, 55, 63]))
    L = k + int(rng.choice([0, 1, 7, 15, 19, 29, 30, 42, 59, 87, 119])) if rng.random() < 0.8 else k + int(rng.integers(0, 130))
    n = int(rng.integers(1, 1800))
    glen = max(L + 1, int(rng.choice([L + 9, 400, 4000, 30000])))
    rc, first_seen = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    prune = first_seen and L > k and rng.random() < 0.4
    route = str(rng.choice(["local", "tiles", "supermers"]))          # (supermers: by packed key, k = 15..31, else the level-by-level route)
    os.environ["KATOME_DIST_ROUTE"] = route
    if world == 1:
        os.environ["KATOME_FORCE_SHARDED"] = "1"
    else:
        os.environ.pop("KATOME_FORCE_SHARDED", None)
    reads = o.synth_reads(int(rng.integers(0, 1000)), n, L, glen, float(rng.choice([0.0, 1e-3, 2e-2])), int(rng.choice([0, 0, 3])))
    has_n = (reads == ord("N")).any(axis=1)
    clean = reads.copy(); clean[clean == ord("N")] = ord("G")
    what = dict(case=c, world=world, k=k, L=L, n=n, glen=glen, rc=rc, first_seen=first_seen, prune=prune, route=route)
    print("case", what, file=sys.stderr, flush=True)                 # (a case that never returns is the last one named)
    try:
        g, rb = GpuGraph.create_from_packed(pack_reads_ascii(clean).reshape(-1).copy(), n, L, skip=has_n.astype(np.uint8), reverse_complement=rc,
                                            k=k, n_devices=world, ranks_share_device=True, first_seen_order=first_seen, remove_dead_paths=prune)
        ref = o.build_ascii(reads, k, rc, remove_dead_paths=prune)
        ok = (g.n_nodes, g.n_edges, rb) == (ref.n_nodes, ref.n_edges, ref.read_bytes)
        if ok and first_seen:
            ok = np.array_equal(g.edge_label, ref.edge_label) and np.array_equal(g.edge_weight, ref.edge_weight) and \
                np.array_equal(g.edge_src, ref.edge_src) and np.array_equal(g.edge_dst, ref.edge_dst)
        elif ok:
            ok = g.multiset() == ref.multiset()
            if ok and g.n_edges:
                ek, nk = g.key_ints("edge"), g.key_ints("node")
                mask = (1 << (2 * (k - 1))) - 1
                ok = len(set(nk)) == len(nk) and all(nk[int(s)] == e >> 2 and nk[int(d)] == e & mask for e, s, d in zip(ek, g.edge_src, g.edge_dst))
        if not ok:
            bad += 1
            print("MISMATCH", what, (g.n_nodes, g.n_edges), (ref.n_nodes, ref.n_edges), flush=True)
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("ERROR", what, type(e).__name__, str(e)[:200], flush=True)
print("%d cases, %d mismatches" % (cases, bad))
