"""Times Shrinkable::shrink in its two forms (katome_dev_shrink_mode) on C3-like reads after the first pruning -- the graph the
reference's collapse() hands to shrink (collapser.rs:31): python tools/time_shrink.py [reads ...]   -> one JSON line per size"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from katome_amd import device as kd
from katome_amd.workloads import WORKLOADS

for reads in [int(a) for a in sys.argv[1:]] or [2_000_000]:
    w = WORKLOADS["c3"].scaled(reads)
    packed, skip = kd.synth_reads(0, w.reads, w.read_len, w.genome_len, w.err_rate, 0, device=0)
    b = kd.Builder(w.k, True, table_slots_hint=int(w.expected_distinct_canonical() * 2.2), first_seen_order=True)
    step = 4 << 20
    for r0 in range(0, w.reads, step):
        b.count_reads(packed, min(step, w.reads - r0), w.read_len, None, first_read=r0)
    dg = b.finalize()
    out = {"reads": reads, "edges": dg.n_edges, "nodes": dg.n_nodes}
    del dg
    dg, st = b.remove_dead_paths()
    out["edges_after_pruning"], out["nodes_after_pruning"] = dg.n_edges, dg.n_nodes
    del dg
    for mode in ("fast", "exact"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dc = b.shrink(mode)
        torch.cuda.synchronize()
        out["shrink_%s_ms" % mode] = (time.perf_counter() - t0) * 1e3
        out["edges_after_shrink_%s" % mode] = dc.n_edges
        if mode == "exact":
            out["shrink_exact_host_ms"] = b.last_shrink_host_ms
        del dc
    b.close()
    print(json.dumps(out), flush=True)
