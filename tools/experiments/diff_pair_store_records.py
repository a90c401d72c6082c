"""Forensics for the lost-tile finding (profiles/r03_pair_store.md): the records that expand_tiles_kernel<.., TO_TABLE=false>
wrote in a 4-rank first-seen build, dumped by the diagnostic builds d0 (shipped kernel) and d1 (the bad shape v1)
(tools/make_pair_store_variants.py dump; KATOME_DUMP_RECORDS), compared as multisets: which records does d1 get wrong, and how
do they relate to the ones it is missing?  One process per library."""
import glob
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(reads):
    sys.path.insert(0, ROOT)
    import torch
    from katome_amd import device as kd
    from katome_amd.build import GpuGraph
    packed, _ = kd.synth_reads(0, reads, 150, reads, 1e-3, 0)
    host = packed[:reads * 38].cpu().numpy().copy()
    del packed
    torch.cuda.empty_cache()
    g, _ = GpuGraph.create_from_packed(host, reads, 150, reverse_complement=True, k=31, first_seen_order=True, n_devices=4, ranks_share_device=True)
    print(json.dumps({"lib": os.environ.get("KATOME_LIB"), "edges": int(g.n_edges)}), flush=True)


def load(prefix):
    """{(k, stride): structured array of (key words..., w, a, b)} over all calls"""
    out = {}
    for path in sorted(glob.glob(prefix + ".*.bin")):
        with open(path, "rb") as f:
            head = np.frombuffer(f.read(64), dtype="<u8")
            n, nwk, k, span, stride, has_seen = (int(x) for x in head[:6])
            keys = np.frombuffer(f.read(8 * n * nwk), dtype="<u8").reshape(n, nwk)
            w = np.frombuffer(f.read(4 * n), dtype="<u4")
            seen = np.frombuffer(f.read(16 * n), dtype="<u8").reshape(n, 2) if has_seen else np.zeros((n, 2), "<u8")
        cols = [keys[:, q] for q in range(nwk)] + [w.astype("<u8"), seen[:, 0], seen[:, 1]]
        out.setdefault((k, stride, span, nwk), []).append(np.stack(cols, axis=1))
    return {key: np.concatenate(v) for key, v in out.items()}


def rows_sorted(a):
    order = np.lexsort(tuple(a[:, c] for c in range(a.shape[1] - 1, -1, -1)))
    return a[order]


def multiset_diff(a, b):
    """rows of a not matched in b, rows of b not matched in a (multisets)"""
    a, b = rows_sorted(a), rows_sorted(b)
    va = np.ascontiguousarray(a).view([("", a.dtype)] * a.shape[1]).ravel()
    vb = np.ascontiguousarray(b).view([("", b.dtype)] * b.shape[1]).ravel()
    ua, ca = np.unique(va, return_counts=True)
    ub, cb = np.unique(vb, return_counts=True)
    only_a = np.setdiff1d(ua, ub)
    only_b = np.setdiff1d(ub, ua)
    common, ia, ib = np.intersect1d(ua, ub, return_indices=True)
    diff_counts = int((ca[ia] != cb[ib]).sum())
    return only_a.view(a.dtype).reshape(-1, a.shape[1]), only_b.view(b.dtype).reshape(-1, b.shape[1]), diff_counts


def kmer(words, k):
    v = 0
    for x in words:
        v = (v << 64) | int(x)
    return "".join("ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    tmp = os.environ.get("TMPDIR", "/tmp")
    for name in ("d0", "d1"):
        env = dict(os.environ, KATOME_DIST_ROUTE="tiles", KATOME_LIB=os.path.join(ROOT, "build_variants", "libkatome_gpu_%s.so" % name),
                   KATOME_DUMP_RECORDS=os.path.join(tmp, "rec_" + name))
        for old in glob.glob(os.path.join(tmp, "rec_%s.*.bin" % name)):
            os.remove(old)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(reads)], env=env, timeout=900)
        if r.returncode:
            print("build with", name, "failed:", r.returncode)
            return 1
    good, bad = load(os.path.join(tmp, "rec_d0")), load(os.path.join(tmp, "rec_d1"))
    for shape in sorted(good):
        k, stride, span, nwk = shape
        g, b = good[shape], bad.get(shape)
        if b is None:
            print(shape, "missing from d1")
            continue
        missing, wrong, diff_counts = multiset_diff(g, b)
        print("level k=%d stride=%d span=%d words=%d: %d records (d0) %d (d1); in d0 only %d, in d1 only %d, multiplicity differs %d"
              % (k, stride, span, nwk, len(g), len(b), len(missing), len(wrong), diff_counts), flush=True)
        if not len(wrong):
            continue
        # how do the wrong records relate to the missing ones?  match on (w, a, b): same pair and weight, another key
        def index(rows):
            d = {}
            for row in rows:
                d.setdefault(tuple(int(x) for x in row[nwk:]), []).append(row[:nwk])
            return d
        mi = index(missing[:200000])
        same_payload = 0
        shown = 0
        xor_hist = {}
        for row in wrong[:200000]:
            cand = mi.get(tuple(int(x) for x in row[nwk:]))
            if not cand:
                continue
            same_payload += 1
            x = [int(a) ^ int(c) for a, c in zip(row[:nwk], cand[0])]
            xor_hist[tuple(x)] = xor_hist.get(tuple(x), 0) + 1
            if shown < 12:
                shown += 1
                print("   wrong %s  for  %s   (w=%d a=%d b=%d)" % (kmer(row[:nwk], k), kmer(cand[0], k), int(row[nwk]), int(row[nwk + 1]), int(row[nwk + 2])))
        print("   wrong records whose (w, a, b) equals a missing record's: %d of %d looked at" % (same_payload, min(len(wrong), 200000)))
        if nwk == 1:
            # is the wrong key the right key shifted right by some number of bits?
            shifts = {}
            n_show = 0
            for row in wrong[:200000]:
                cand = mi.get(tuple(int(x) for x in row[nwk:]))
                if not cand:
                    continue
                bad_key, right = int(row[0]), int(cand[0][0])
                sh = [t for t in range(1, 64) if (right >> t) == bad_key]
                key = sh[0] if sh else -1
                shifts[key] = shifts.get(key, 0) + 1
                if n_show < 40:
                    n_show += 1
                    print("      bad %016x right %016x shift %s w=%d a=%d b=%d (a %% 240 = %d)" % (bad_key, right, sh[:3], int(row[1]), int(row[2]), int(row[3]), int(row[2]) % 240))
            print("   histogram of the shift (-1: not a shift):", sorted(shifts.items()))
        # is a wrong key the key of ANOTHER correct record (a neighbour's key stored in this record's place)?
        gk = np.ascontiguousarray(g[:, :nwk]).view([("", g.dtype)] * nwk).ravel()
        wk = np.ascontiguousarray(wrong[:, :nwk]).view([("", g.dtype)] * nwk).ravel()
        print("   wrong keys that are keys of some correct record of this level: %d of %d" % (int(np.isin(wk, gk).sum()), len(wk)))
        # payload-only differences: same key, other pair
        mk = np.ascontiguousarray(missing[:, :nwk]).view([("", g.dtype)] * nwk).ravel()
        print("   wrong records whose KEY is among the missing records' keys (payload differs): %d" % int(np.isin(wk, mk).sum()))
        for row in wrong[:6]:
            print("   e.g. d1-only: %s w=%d a=%d b=%d" % (kmer(row[:nwk], k), int(row[nwk]), int(row[nwk + 1]), int(row[nwk + 2])))
        for row in missing[:6]:
            print("   e.g. d0-only: %s w=%d a=%d b=%d" % (kmer(row[:nwk], k), int(row[nwk]), int(row[nwk + 1]), int(row[nwk + 2])))
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        sys.exit(main())
