"""Random small FASTQ files of unequal read lengths through the host entry (katome_build_files) against the oracle:
plain and reference numbering, with and without remove_dead_paths, small batches.  usage: python tools/fuzz_files.py [cases=100] [seed=0]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o
from katome_amd.build import GpuGraph, InputFileType, set_global_k_sizes

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
tmp = tempfile.mkdtemp()
for c in range(cases):
    k = int(rng.choice([3, 5, 11, 16, 21, 31, 32, 33, 40, 47, 55, 63]))
    n = int(rng.integers(1, 1200))
    glen = int(rng.choice([5 * k + 20, 2000, 20000]))
    genome = rng.integers(0, 4, glen)
    lines = []
    for i in range(n):
        ln = int(rng.integers(k, min(glen, 6 * k + 10)))
        s0 = int(rng.integers(0, glen - ln + 1))
        r = genome[s0:s0 + ln].copy()
        m = rng.random(ln) < float(rng.choice([0.0, 0.01]))
        r[m] = rng.integers(0, 4, int(m.sum()))
        s = "".join("ACGT"[x] for x in r)
        if rng.random() < 0.03:
            p = int(rng.integers(0, ln)); s = s[:p] + "N" + s[p + 1:]
        lines += ["@r%d" % i, s, "+", "I" * ln]
    path = os.path.join(tmp, "f%d.fq" % c)
    open(path, "w").write("\n".join(lines) + "\n")
    rc, first_seen = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    prune = first_seen and rng.random() < 0.5
    if rng.random() < 0.5:
        os.environ["KATOME_VAR_BATCH_RECORDS"] = str(int(rng.choice([200, 3000])))
    else:
        os.environ.pop("KATOME_VAR_BATCH_RECORDS", None)
    set_global_k_sizes(k)
    g, rb = GpuGraph.create([path], InputFileType.Fastq, rc, 0, first_seen_order=first_seen, remove_dead_paths=prune)
    ref = o.build_files([path], k, rc, remove_dead_paths=prune)
    ok = rb == ref.read_bytes and (g.n_nodes, g.n_edges) == (ref.n_nodes, ref.n_edges)
    if ok and first_seen:
        ok = (np.array_equal(g.edge_label, ref.edge_label) and np.array_equal(g.edge_weight, ref.edge_weight) and
              np.array_equal(g.edge_src, ref.edge_src) and np.array_equal(g.edge_dst, ref.edge_dst))
    elif ok:
        ok = g.multiset() == ref.multiset()
    os.remove(path)
    if not ok:
        bad += 1
        print("MISMATCH case %d: k=%d n=%d glen=%d rc=%s first_seen=%s prune=%s batch=%s" % (c, k, n, glen, rc, first_seen, prune, os.environ.get("KATOME_VAR_BATCH_RECORDS")), flush=True)
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
