import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import pack_reads_ascii
from oracle import oracle as o
from katome_amd import device as kd
for k, L, n in ((5, 50, 300), (31, 150, 2000), (11, 60, 500)):
    reads = o.synth_reads(1, n, L, 3000, 1e-2, 0)
    packed = torch.from_numpy(pack_reads_ascii(reads).reshape(-1).copy()).cuda()
    b = kd.Builder(k, True, table_slots_hint=1 << 12)
    b.count_reads(packed, n, L, None, first_read=0)
    ek, ew = b.edges()
    ref = o.build_ascii(reads, k, True)
    print(k, "edges", ek.shape[0], ref.n_edges, "weight sum", int(ew.to(torch.int64).sum()), int(ref.edge_weight.astype(np.int64).sum()), "max", int(ew.max()), flush=True)
    b.close()
