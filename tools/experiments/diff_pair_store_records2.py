"""Second look at the wrong records of tools/diff_pair_store_records.py (run that first: it leaves the dumps in $TMPDIR):
where in the output do they sit (record index), and how does the shift depend on the sub-window?"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from diff_pair_store_records import load

tmp = os.environ.get("TMPDIR", "/tmp")
good = load(os.path.join(tmp, "rec_d0"))
shape = [s for s in good if s[1] == 1][0]
g = good[shape]
gv = np.ascontiguousarray(g).view([("", g.dtype)] * g.shape[1]).ravel()
gset = np.unique(gv)
by_payload = {}
for row in g:
    by_payload.setdefault((int(row[1]), int(row[2]), int(row[3])), int(row[0]))
rows = []
for path in sorted(glob.glob(os.path.join(tmp, "rec_d1.*.bin"))):
    with open(path, "rb") as f:
        head = np.frombuffer(f.read(64), dtype="<u8")
        n, nwk, k, span, stride, has_seen = (int(x) for x in head[:6])
        if stride != 1:
            continue
        keys = np.frombuffer(f.read(8 * n * nwk), dtype="<u8").reshape(n, nwk)
        w = np.frombuffer(f.read(4 * n), dtype="<u4")
        seen = np.frombuffer(f.read(16 * n), dtype="<u8").reshape(n, 2)
    a = np.stack([keys[:, 0], w.astype("<u8"), seen[:, 0], seen[:, 1]], axis=1)
    av = np.ascontiguousarray(a).view([("", a.dtype)] * 4).ravel()
    bad_idx = np.nonzero(~np.isin(av, gset))[0]
    print(os.path.basename(path), "records", n, "wrong", len(bad_idx), flush=True)
    runs = np.split(bad_idx, np.nonzero(np.diff(bad_idx) != 1)[0] + 1) if len(bad_idx) else []
    print("   runs of consecutive wrong records: lengths", sorted(set(len(r) for r in runs)), " starts mod 6:", sorted(set(int(r[0]) % 6 for r in runs)),
          " starts mod 64:", sorted(set(int(r[0]) % 64 for r in runs))[:70])
    shown = 0
    for r in runs[:400]:
        for i in r:
            bad = int(keys[i, 0]); right = by_payload.get((int(w[i]), int(seen[i, 0]), int(seen[i, 1])))
            if right is None:
                continue
            sh = [t for t in range(1, 64) if (right >> t) == bad]
            fwd_first = int(seen[i, 0]) % 240 < 120
            rows.append((int(i), int(i) % 6, int(i) % 64, sh[0] if sh else -1, fwd_first))
            if shown < 60:
                shown += 1
                print("      i=%d i%%6=%d i%%64=%d i%%256=%d bad=%016x right=%016x shift=%s a%%240=%d b%%240=%d w=%d" % (
                    i, i % 6, i % 64, i % 256, bad, right, sh[:2], int(seen[i, 0]) % 240, int(seen[i, 1]) % 240, int(w[i])))
        if shown < 60 and len(r):
            lo, hi = max(int(r[0]) - 2, 0), min(int(r[-1]) + 3, n)
            print("      neighbours %d..%d keys:" % (lo, hi - 1), " ".join("%016x" % int(x) for x in keys[lo:hi, 0]))
tab = {}
for i, m6, m64, sh, fwd in rows:
    tab.setdefault((m6, fwd), {}).setdefault(sh, 0)
    tab[(m6, fwd)][sh] += 1
for key in sorted(tab):
    print("i%%6=%d unflipped=%s: shifts %s" % (key[0], key[1], sorted(tab[key].items())))
