"""A sharded build (thread ranks sharing the card, both routes) against the one-GPU build of the same reads, at a size where
the routes run in slices and the last level is counted by sorting on its own (>= 4 M records): edge multiset by an
order-free checksum, node set, and every edge's end points.  usage: python tools/check_sharded_scale.py [reads=8000000] [world=4] [routes=tiles,local] [table_slots_hint=0] [k=31] [fs]
(fs: the reference's numbering + remove_dead_paths on the gathered graph; every array must equal the one-GPU build's)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from katome_amd import device as kd  # noqa: E402
from katome_amd.build import GpuGraph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 150
k = int(sys.argv[5]) if len(sys.argv) > 5 else 31
fs = len(sys.argv) > 6 and sys.argv[6] == "fs"
hint = int(sys.argv[4]) if len(sys.argv) > 4 else 0
packed_d, _ = kd.synth_reads(0, n, L, n // 2, 1e-3, 0, device=0)
packed = packed_d.cpu().numpy()
del packed_d
torch.cuda.empty_cache()
M = np.uint64(0x9E3779B97F4A7C15)


def digest(g):
    nw = 1 if k <= 31 else 2
    keys = g.edge_key.reshape(-1, nw)
    key = keys[:, 0] if nw == 1 else (keys[:, 0] * np.uint64(0xD6E8FEB86659FD93)) ^ keys[:, 1]         # (two words folded into one for the checksums)
    h = (key * M) ^ (key >> np.uint64(29))
    s = int((h * g.edge_weight.astype(np.uint64)).sum(dtype=np.uint64))
    x = int(np.bitwise_xor.reduce(h + g.edge_weight.astype(np.uint64)))
    nk = g.node_key.reshape(-1, nw)
    nf = nk.reshape(-1)
    ns = int(((nf * M) ^ (nf >> np.uint64(31))).sum(dtype=np.uint64))
    src, dst = g.edge_src.astype(np.int64), g.edge_dst.astype(np.int64)
    if nw == 1:
        mask = np.uint64((1 << (2 * (k - 1))) - 1)
        ok_src = bool((nk[src, 0] == (keys[:, 0] >> np.uint64(2))).all())
        ok_dst = bool((nk[dst, 0] == (keys[:, 0] & mask)).all())
    else:                         # 128-bit keys: source = key >> 2, target = low 2(k-1) bits
        hi, lo = keys[:, 0], keys[:, 1]
        s_hi, s_lo = hi >> np.uint64(2), (lo >> np.uint64(2)) | (hi << np.uint64(62))
        bits = 2 * (k - 1)
        t_hi = hi & np.uint64((1 << (bits - 64)) - 1) if bits > 64 else np.zeros_like(hi)
        t_lo = lo if bits >= 64 else lo & np.uint64((1 << bits) - 1)
        ok_src = bool(((nk[src, 0] == s_hi) & (nk[src, 1] == s_lo)).all())
        ok_dst = bool(((nk[dst, 0] == t_hi) & (nk[dst, 1] == t_lo)).all())
    return (g.n_nodes, g.n_edges, s, x, ns, int(g.edge_weight.sum(dtype=np.uint64))), ok_src and ok_dst


if fs:
    import hashlib

    def arrays(g):
        h = hashlib.sha256()
        for a in (g.edge_key, g.edge_weight, g.edge_src, g.edge_dst, g.node_key, g.edge_label):
            h.update(np.ascontiguousarray(a).tobytes())
        return g.n_nodes, g.n_edges, h.hexdigest()
    t0 = time.time()
    one, _ = GpuGraph.create_from_packed(packed, n, L, reverse_complement=True, k=k, first_seen_order=True, remove_dead_paths=True)
    want = arrays(one)
    print("one GPU  ", want, "%.1f s" % (time.time() - t0), flush=True)
    del one
    bad = False
    for route in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("tiles", "local")):
        os.environ["KATOME_DIST_ROUTE"] = route
        t0 = time.time()
        g, _ = GpuGraph.create_from_packed(packed, n, L, reverse_complement=True, k=k, n_devices=world, ranks_share_device=os.environ.get("KATOME_REAL_DEVICES") != "1",
                                            first_seen_order=True, remove_dead_paths=True, table_slots_hint=hint)
        got = arrays(g)
        print("%d ranks %s" % (world, route), got, "same:", got == want, "%.1f s" % (time.time() - t0), flush=True)
        bad |= got != want
        del g
    print("OK" if not bad else "MISMATCH")
    sys.exit(1 if bad else 0)
t0 = time.time()
one, _ = GpuGraph.create_from_packed(packed, n, L, reverse_complement=True, k=k)
want, ok = digest(one)
print("one GPU  ", want, "end points ok:", ok, "%.1f s" % (time.time() - t0), flush=True)
del one
bad = not ok
for route in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("tiles", "local")):
    os.environ["KATOME_DIST_ROUTE"] = route
    t0 = time.time()
    g, _ = GpuGraph.create_from_packed(packed, n, L, reverse_complement=True, k=k, n_devices=world, ranks_share_device=os.environ.get("KATOME_REAL_DEVICES") != "1",
                                        table_slots_hint=hint)
    got, ok = digest(g)
    print("%d ranks %s" % (world, route), got, "end points ok:", ok, "same:", got == want, "%.1f s" % (time.time() - t0), flush=True)
    bad |= (got != want) or not ok
    del g
print("OK" if not bad else "MISMATCH")
sys.exit(1 if bad else 0)
