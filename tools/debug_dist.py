import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from katome_amd import device as kd, dist as kdist
from katome_amd.workloads import WORKLOADS
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
wl = WORKLOADS["c3"].scaled(R)
packed, skip = kd.synth_reads(0, wl.reads, wl.read_len, wl.genome_len, wl.err_rate, 0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 20)
def edges_single():
    b = kd.Builder(wl.k, True)
    span = b.tile_span(wl.read_len)
    for r0 in range(0, wl.reads, B):
        nr = min(B, wl.reads - r0)
        b.insert_tiles(b.extract_tiles(packed, nr, wl.read_len, span, None, first_read=r0), span)
    k, w = b.edges(); k, w = k.clone(), w.clone(); b.close(); return k, w
def edges_variant(use_partition, use_exchange, use_expand_records):
    b = kd.Builder(wl.k, True)
    span = b.tile_span(wl.read_len); nwt = b.tile_words(span)
    for r0 in range(0, wl.reads, B):
        nr = min(B, wl.reads - r0)
        rec = b.extract_tiles(packed, nr, wl.read_len, span, None, first_read=r0)
        if use_partition:
            rec, counts = b.partition(rec, 1, key_words=nwt)
            rec = rec[:counts[0] * nwt]
        if use_exchange:
            rec, _ = kdist._exchange(rec, [rec.numel() // nwt], nwt, None)
        b.insert_tiles(rec, span)
    if use_expand_records:
        keys, weights = b.expand_tiles()
        if use_partition:
            keys, counts, weights = b.partition(keys, 1, key_words=b.nw, values=weights)
        if use_exchange:
            n = weights.numel()
            keys, _ = kdist._exchange(keys, [n], b.nw, None)
            weights, _ = kdist._exchange(weights, [n], 1, None)
        b.insert(keys, weights)
    k, w = b.edges(); k, w = k.clone(), w.clone(); b.close(); return k, w
k0, w0 = edges_single()
print("single", k0.shape[0], int(w0.to(torch.int64).sum()))
for flags in [(False, False, True), (True, False, False), (True, False, True), (False, True, False), (True, True, True)]:
    k1, w1 = edges_variant(*flags)
    same = k1.shape == k0.shape and bool(torch.equal(k1, k0)) and bool(torch.equal(w1, w0))
    print("partition=%s exchange=%s expand_records=%s ->" % flags, k1.shape[0], int(w1.to(torch.int64).sum()), "SAME" if same else "DIFFERENT", flush=True)
dist.destroy_process_group()
