"""One-off confidence run at a size between the unit tests and the benchmarks: every stage of assemble_with_graph before
collapse on the device, index for index against the oracle (the oracle is test infrastructure; this tool is a checker).
usage: python tools/soak_pipeline.py [reads=300000] [k=31] [genome=300000] [threshold=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from helpers import pack_reads_ascii
from oracle import oracle as o
from katome_amd import device as kd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
glen = int(sys.argv[3]) if len(sys.argv) > 3 else 300_000
thr = int(sys.argv[4]) if len(sys.argv) > 4 else 2
L, rc = 150, True
t = time.time()
reads = o.synth_reads(0, n, L, glen, 2e-3, 0)
packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
print("reads ready in %.1f s" % (time.time() - t), flush=True)
b = kd.Builder(k, rc, first_seen_order=True)
step = 1 << 16
for r0 in range(0, n, step):
    b.count_reads(packed, min(step, n - r0), L, None, first_read=r0)
dg = b.finalize()
o.set_genome_length(glen)


def same(dg, ref, what):
    ok = (dg.n_nodes, dg.n_edges) == (ref.n_nodes, ref.n_edges)
    ok = ok and np.array_equal(dg.edge_label.cpu().numpy(), ref.edge_label)
    ok = ok and np.array_equal(dg.edge_weight.cpu().numpy().view(np.uint32), ref.edge_weight)
    ok = ok and np.array_equal(dg.edge_src.cpu().numpy().view(np.uint64), ref.edge_src)
    ok = ok and np.array_equal(dg.edge_dst.cpu().numpy().view(np.uint64), ref.edge_dst)
    if dg.edge_age is not None:
        ok = ok and np.array_equal(dg.edge_age.cpu().numpy().view(np.uint32).astype(np.uint64) + 1, ref.edge_slot)
    print("%-8s %10d nodes %10d edges  %s" % (what, dg.n_nodes, dg.n_edges, "SAME" if ok else "DIFFERENT"), flush=True)
    return ok


def ref(stages):
    t = time.time()
    r = o.build_ascii(reads, k, rc, remove_weak_edges=thr, stages=stages)
    print("  (oracle '%s': %.1f s)" % (stages, time.time() - t), flush=True)
    return r


good = same(dg, ref(""), "build")
dg, st = b.remove_dead_paths()
print("  passes %d, removed %d edges (%d by repeated indices), %d nodes, host_ms %.1f" % (st["passes"], st["removed_edges"], st["removed_by_duplicates"], st["removed_nodes"], st["host_ms"]))
good &= same(dg, ref("d"), "d")
b.standardize_contigs(); b.remove_weak_edges(thr); b.standardize_contigs()
good &= same(b.graph(), ref("dcwc"), "dcwc")
b.standardize_edges(glen, thr)
dg, st = b.remove_dead_paths()
good &= same(dg, ref("dcwced"), "dcwced")
b.close()
print("ALL SAME" if good else "MISMATCH")
sys.exit(0 if good else 1)
