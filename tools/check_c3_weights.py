"""One build of N C3-like reads by packed key: edge count, weight sum against 2 * reads * windows, strictly ascending keys, table counts
(which level was counted where).  usage: python tools/check_c3_weights.py [reads=60000000]; KATOME_SORTED_COUNT=0 for the table path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from katome_amd import device as kd
from katome_amd.workloads import WORKLOADS
wl = WORKLOADS["c3"].scaled(int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000)
packed, _ = kd.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, 0)
b = kd.Builder(wl.k, True, table_slots_hint=int(wl.expected_distinct_canonical() * 1.8))
step = 16 << 20
for r0 in range(0, wl.reads, step):
    b.count_reads(packed, min(step, wl.reads - r0), wl.read_len, None, first_read=r0)
ek, ew = b.edges()
tot = 0
for i in range(0, ew.numel(), 1 << 27):
    tot += int(ew[i:i + (1 << 27)].to(torch.int64).sum().item())
print("edges", ek.shape[0], "weight sum", tot, "expected", 2 * wl.reads * wl.windows_per_read, "max", int(ew.max()), "sorted keys", bool((ek[1:, 0] > ek[:-1, 0]).all()))
print(b.counts())
