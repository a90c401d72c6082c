"""Every stage of assemble_with_graph before collapse on one GPU, at scale: first-seen-order build, remove_dead_paths,
standardize_contigs, remove_weak_edges(t), standardize_contigs, standardize_edges(G, k, t), remove_dead_paths.
usage: python tools/bench_pipeline.py [--workload c3] [--reads N] [--threshold 2]"""
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
    ap.add_argument("--threshold", type=int, default=2)
    a = ap.parse_args()
    wl = workloads.WORKLOADS[a.workload]
    if a.reads:
        wl = wl.scaled(a.reads)
    packed, skip = kd.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, wl.n_inject_percent, device=0)
    skip_arg = skip if wl.n_inject_percent else None
    b = kd.Builder(wl.k, wl.reverse_complement, table_slots_hint=int(wl.expected_distinct_canonical() * 2.2), first_seen_order=True)
    out = {"workload": wl.name, "reads": wl.reads, "k": wl.k, "threshold": a.threshold, "stages": []}

    def stage(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        g = b.graph()
        out["stages"].append({"stage": name, "ms": (time.perf_counter() - t0) * 1e3, "nodes": g.n_nodes, "edges": g.n_edges})
        print(out["stages"][-1], file=sys.stderr, flush=True)

    def build():
        for r0 in range(0, wl.reads, 4 << 20):
            b.count_reads(packed, min(4 << 20, wl.reads - r0), wl.read_len, skip_arg, first_read=r0)
        b.finalize()
    stage("build (first-seen order)", build)
    stage("remove_dead_paths", b.remove_dead_paths)
    stage("standardize_contigs", b.standardize_contigs)
    stage("remove_weak_edges", lambda: b.remove_weak_edges(a.threshold))
    stage("standardize_contigs", b.standardize_contigs)
    stage("standardize_edges", lambda: b.standardize_edges(wl.genome_len, a.threshold))
    stage("remove_dead_paths", b.remove_dead_paths)
    out["total_ms"] = sum(s["ms"] for s in out["stages"])
    b.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
