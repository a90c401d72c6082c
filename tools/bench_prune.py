"""remove_dead_paths and shrink at scale: build a synthetic workload in first-seen order on one GPU, prune, shrink, print the stats.
usage: python tools/bench_prune.py [--workload c3] [--reads N] [--cpu-reads M]
With --cpu-reads the oracle (1 core) prunes the first M reads' graph for a CPU figure beside it."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from katome_amd import device as kd  # noqa: E402
from katome_amd import workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--cpu-reads", type=int, default=0)
    a = ap.parse_args()
    wl = workloads.WORKLOADS[a.workload]
    if a.reads:
        wl = wl.scaled(a.reads)
    torch.cuda.set_device(0)
    packed, skip = kd.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, wl.n_inject_percent, device=0)
    skip_arg = skip if wl.n_inject_percent else None
    b = kd.Builder(wl.k, wl.reverse_complement, table_slots_hint=int(wl.expected_distinct_canonical() * 2.2),
                   first_seen_order=True)
    b.profile(True)
    span = b.tile_span(wl.read_len)
    batch = 4 << 20
    t0 = time.perf_counter()
    for r0 in range(0, wl.reads, batch):
        nr = min(batch, wl.reads - r0)
        if span > 1:
            b.insert_tiles(b.extract_tiles(packed, nr, wl.read_len, span, skip_arg, first_read=r0), span)
        else:
            b.insert(b.extract_fixed(packed, nr, wl.read_len, skip_arg, first_read=r0))
    dg = b.finalize()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    before = (dg.n_nodes, dg.n_edges)
    t0 = time.perf_counter()
    dg, st = b.remove_dead_paths()
    torch.cuda.synchronize()
    t_prune = time.perf_counter() - t0
    t0 = time.perf_counter()
    dc = b.shrink()
    torch.cuda.synchronize()
    t_shrink = time.perf_counter() - t0
    kd.release_cache(0)                 # the library's cached blocks: torch needs room for the statistics below
    kmers = dc.edge_kmers
    shrink = {"ms": t_shrink * 1e3, "nodes": dc.n_nodes, "edges": dc.n_edges, "label_bytes": dc.label_bytes,
              "longest_path_kmers": int(kmers.max().item()) if dc.n_edges else 0,
              "mean_path_kmers": float(kmers.double().mean().item()) if dc.n_edges else 0.0}
    out = {"workload": wl.name, "shrink": shrink, "reads": wl.reads, "k": wl.k, "rc": wl.reverse_complement, "build_ms": t_build * 1e3,
           "prune_ms": t_prune * 1e3, "before": before, "after": (dg.n_nodes, dg.n_edges), "stats": st,
           "phases": {k: v for k, v in b.profile_read().items() if v[1]}}
    b.close()
    if a.cpu_reads:
        from oracle import oracle as o
        n = min(a.cpu_reads, wl.reads)
        reads = o.synth_reads(0, n, wl.read_len, wl.genome_len, wl.err_rate, wl.n_inject_percent)
        t0 = time.perf_counter()
        full = o.build_ascii(reads, wl.k, wl.reverse_complement)
        t1 = time.perf_counter()
        pruned = o.build_ascii(reads, wl.k, wl.reverse_complement, remove_dead_paths=True)
        t2 = time.perf_counter()
        shrunk = o.build_ascii(reads, wl.k, wl.reverse_complement, stages="ds")
        t3 = time.perf_counter()
        out["cpu"] = {"reads": n, "edges_before": full.n_edges, "edges_after": pruned.n_edges,
                      "prune_s": (t2 - t1) - (t1 - t0), "passes": o.last_prune_passes(),
                      "shrink_s": (t3 - t2) - (t2 - t1), "edges_shrunk": shrunk.n_edges}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
