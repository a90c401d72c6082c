"""How many tiles and k-mer records each rank of a sharded build holds (KATOME_DIST_STATS=1 prints them): 8 M C3-like reads through
1 and 8 thread ranks on one card.  usage: KATOME_DIST_STATS=1 python tools/dist_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from katome_amd import device as kd
from katome_amd.build import GpuGraph
n, G = 8_000_000, 4_000_000
packed, skip = kd.synth_reads(0, n, 150, G, 1e-3, 0, device=0)
host = packed.cpu().numpy()
for world in (1, 8):
    os.environ["KATOME_FORCE_SHARDED"] = "1"
    g, rb = GpuGraph.create_from_packed(host, n, 150, reverse_complement=True, k=31, n_devices=world, ranks_share_device=True)
    print("world", world, "edges", g.n_edges, "nodes", g.n_nodes, flush=True)
