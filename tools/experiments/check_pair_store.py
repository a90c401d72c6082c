"""One confirming run for round 2's lost-tile finding (table.hip, expand_tiles_kernel<.., TO_TABLE=false> with sequence
numbers): the sharded first-seen-order build (4 thread ranks on one card, level-by-level route: both record expansions with
numbers) against the one-GPU build of the same reads, array for array, with the shipped library and with each variant of
tools/make_pair_store_variants.py.  One process per library (KATOME_LIB)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(reads):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from katome_amd import device as kd
    from katome_amd.build import GpuGraph
    L, k = 150, 31
    packed, _ = kd.synth_reads(0, reads, L, reads, 1e-3, 0)
    host = packed[:reads * 38].cpu().numpy().copy()
    del packed
    torch.cuda.empty_cache()
    one, _ = GpuGraph.create_from_packed(host, reads, L, reverse_complement=True, k=k, first_seen_order=True)
    out = {"lib": os.environ.get("KATOME_LIB", "shipped"), "reads": reads, "edges": int(one.n_edges), "ranks": int(os.environ.get("KATOME_CHECK_RANKS", "4"))}
    for rep in range(3):
        g, _ = GpuGraph.create_from_packed(host, reads, L, reverse_complement=True, k=k, first_seen_order=True,
                                           n_devices=int(os.environ.get("KATOME_CHECK_RANKS", "4")), ranks_share_device=True)
        same_counts = (g.n_edges, g.n_nodes) == (one.n_edges, one.n_nodes)
        diff = -1
        if same_counts:
            diff = int((np.asarray(g.edge_label) != np.asarray(one.edge_label)).any(axis=1).sum()
                       + (np.asarray(g.edge_weight) != np.asarray(one.edge_weight)).sum()
                       + (np.asarray(g.edge_src) != np.asarray(one.edge_src)).sum()
                       + (np.asarray(g.edge_dst) != np.asarray(one.edge_dst)).sum())
        out["rep%d" % rep] = {"counts_equal": bool(same_counts), "edges": int(g.n_edges), "entries_that_differ": diff}
    print(json.dumps(out), flush=True)


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    named = len(sys.argv) > 2 and not sys.argv[2].replace(",", "").isdigit()                # e.g. "aA,aC": build_variants/libkatome_gpu_aA.so ...
    only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and not named else None       # e.g. "0,1,5" (0 = the shipped library)
    libs = [os.path.join(ROOT, "build_variants", "libkatome_gpu_%s.so" % v) for v in sys.argv[2].split(",")] if named else [None] + [os.path.join(ROOT, "build_variants", "libkatome_gpu_v%d.so" % v) for v in (1, 2, 3, 4, 5, 7, 8, 9, 10)]
    for i, lib in enumerate(libs):
        if only is not None and i not in only:
            continue
        env = dict(os.environ)
        env["KATOME_DIST_ROUTE"] = "tiles"
        if lib:
            if not os.path.exists(lib):
                print(json.dumps({"lib": lib, "error": "not built"}), flush=True)
                continue
            env["KATOME_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(reads)], env=env, timeout=900)
        if r.returncode != 0:
            print(json.dumps({"lib": lib or "shipped", "error": "exit %d" % r.returncode}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        main()
