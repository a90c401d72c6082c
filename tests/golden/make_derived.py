"""Regenerates tests/golden/derived.json from the CPU oracle (run from the repo root)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as o  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
cases = []
for f in ["data1.txt", "data2.txt", "data3.txt"]:
    for k in (31, 40, 63):
        for rc in (False, True):
            g = o.build_files([os.path.join(here, f)], k, rc)
            cases.append(dict(fixture=f, k=k, rc=rc, counts=[g.n_nodes, g.n_edges],
                              weight_sum=int(g.edge_weight.astype("uint64").sum()),
                              stats={kk: (round(v, 6) if isinstance(v, float) else v) for kk, v in g.stats.items()}))
json.dump({"_comment": "DERIVED goldens (not reference-pinned): produced by oracle/katome_oracle.c after it "
                       "reproduced every constant in pinned.json and compress_kat.json. Script: "
                       "tests/golden/make_derived.py", "cases": cases},
          open(os.path.join(here, "derived.json"), "w"), indent=1)
