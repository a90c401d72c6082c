"""End-to-end file path: FASTQ text -> katome_build_files (parallel host ingest, H2D, GPU build, D2H of the graph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as o   # test infrastructure: only used here to WRITE the synthetic FASTQ
from katome_amd.build import GpuGraph, InputFileType, set_global_k_sizes, ingest_files
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
trim = len(sys.argv) > 2 and sys.argv[2] == "trimmed"       # reads cut to 80..150 bases, as after quality trimming
L, k = 150, 31
path = "/tmp/katome_bench.fq"
t = time.time()
with open(path, "wb") as f:
    q = b"I" * L
    step = 500_000
    for r0 in range(0, n, step):
        reads = o.synth_reads(r0, min(step, n - r0), L, max(n // 2, 1000), 1e-3, 1)
        keep = np.random.default_rng(r0).integers(80, L + 1, reads.shape[0]) if trim else np.full(reads.shape[0], L)
        rows = [b"@r%d\n%s\n+\n%s\n" % (r0 + i, reads[i].tobytes()[:keep[i]], q[:keep[i]]) for i in range(reads.shape[0])]
        f.write(b"".join(rows))
size = os.path.getsize(path)
print("wrote %d reads, %.2f GB in %.1f s" % (n, size / 1e9, time.time() - t), flush=True)
set_global_k_sizes(k)
for it in range(3):
    t = time.time(); r = ingest_files([path], InputFileType.Fastq, k); ti = time.time() - t
    t = time.time(); g, rb = GpuGraph.create([path], InputFileType.Fastq, True, 0); tb = time.time() - t
    print("ingest alone %.2f s (%.2f GB/s incl. numpy copies); build_files end-to-end %.2f s = %.3g k-mers/s (%d edges, %d nodes; incl. D2H + numpy copies of %.2f GB)"
          % (ti, size / 1e9 / ti, tb, r["total_windows"] / tb, g.n_edges, g.n_nodes,
             (g.n_edges * (8 + 8 + 4 + 9 + 8) + g.n_nodes * 8) / 1e9), flush=True)
os.remove(path)
