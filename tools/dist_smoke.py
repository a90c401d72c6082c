"""one sharded build through the C ABI with the ranks sharing a card (debugging aid): python tools/dist_smoke.py [world k L n first_seen]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import pack_reads_ascii  # noqa: E402
from katome_amd.build import GpuGraph  # noqa: E402
from oracle import oracle as o  # noqa: E402

world, k, L, n, fs = (int(x) for x in (sys.argv[1:6] + ["2", "11", "50", "260", "0"][len(sys.argv) - 1:]))
reads = o.synth_reads(0, n, L, 3000, 2e-2, 4)
has_n = (reads == ord("N")).any(axis=1)
clean = reads.copy()
clean[clean == ord("N")] = ord("A")
g, rb = GpuGraph.create_from_packed(pack_reads_ascii(clean).reshape(-1).copy(), n, L, skip=has_n.astype(np.uint8), reverse_complement=True,
                                    k=k, n_devices=world, ranks_share_device=True, first_seen_order=bool(fs))
ref = o.build_ascii(reads, k, True)
print("gpu", g.n_nodes, g.n_edges, "oracle", ref.n_nodes, ref.n_edges, "multiset equal:", g.multiset() == ref.multiset())
if g.multiset() != ref.multiset():
    from collections import Counter
    a, b = Counter(g.multiset()), Counter(ref.multiset())
    extra, missing = list((a - b).elements()), list((b - a).elements())
    print("extra", len(extra), extra[:6])
    print("missing", len(missing), missing[:6])
    if fs:
        import numpy as np
        bad = np.nonzero((g.edge_label != ref.edge_label).any(axis=1))[0] if g.n_edges == ref.n_edges else []
        print("positions with another label:", len(bad), list(bad[:10]))
