"""Random small builds through shrink (unitig compaction) against the oracle's literal shrink.  The EXACT form (first-seen-order
builders; optionally after remove_dead_paths): every array index for index, on ANY graph -- tangles, cycles, parts only the
traversal's restarts reach.  The traversal-free form: the multiset of merged edges on the graphs where the reference's result does
not depend on its traversal order (every vertex reachable from a vertex without incoming edges).
usage: python tools/fuzz_shrink.py [cases=200] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import pack_reads_ascii
from oracle import oracle as o
from katome_amd import device as kd
from test_gpu_shrink import _contigs, _reaches_everything_from_inputs

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = compared = 0
for c in range(cases):
    k = int(rng.choice([5, 9, 15, 16, 21, 31, 32, 33, 40, 55, 63]))
    L = k + int(rng.integers(1, 100))
    n = int(rng.integers(1, 2500))
    glen = int(rng.choice([L + 30, 2000, 30000]))
    rc = bool(rng.integers(0, 2))
    err = float(rng.choice([0.0, 0.0, 5e-3, 2e-2]))
    reads = o.synth_reads(int(rng.integers(0, 1000)), n, L, glen, err, 0)
    full = o.build_ascii(reads, k, rc)
    exact = bool(rng.integers(0, 2))
    prune = exact and bool(rng.integers(0, 2))
    if not exact and not _reaches_everything_from_inputs(full):
        continue
    packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
    b = kd.Builder(k, rc, first_seen_order=exact or bool(rng.integers(0, 2)))
    try:
        b.count_reads(packed, n, L)
        b.finalize()
        if prune:
            b.remove_dead_paths()
        want = o.build_ascii(reads, k, rc, stages="ds" if prune else "s")
        if exact and want.n_edges:
            dc = b.shrink("exact")
            ok = ((dc.n_nodes, dc.n_edges) == (want.n_nodes, want.n_edges) and dc.edge_src.cpu().tolist() == want.edge_src.tolist()
                  and dc.edge_dst.cpu().tolist() == want.edge_dst.tolist()
                  and dc.edge_weight.cpu().numpy().view(np.uint32).tolist() == want.edge_weight.tolist() and dc.sequences() == want.edge_seq)
        elif exact:
            ok = True
        else:
            dc = b.shrink("fast")
            ok = (dc.n_nodes, dc.n_edges) == (want.n_nodes, want.n_edges) and _contigs(dc, k) == want.contigs()
    except AssertionError:
        ok = False
    finally:
        b.close()
    compared += 1
    if not ok:
        bad += 1
        print("MISMATCH case %d: k=%d L=%d n=%d glen=%d rc=%s err=%g exact=%s prune=%s" % (c, k, L, n, glen, rc, err, exact, prune), flush=True)
print("%d cases, %d compared, %d mismatches" % (cases, compared, bad))
sys.exit(1 if bad else 0)
