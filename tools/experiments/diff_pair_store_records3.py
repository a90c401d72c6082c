"""Third look (see diff_pair_store_records.py): the diagnostic build d2 stores p | blockIdx << 16 in place of the weight, so a wrong
record tells which thread wrote it and in which trip of its loop.  Single rank through the sharded route."""
import glob, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def read_dump(path):
    with open(path, "rb") as f:
        head = np.frombuffer(f.read(64), dtype="<u8")
        n, nwk, k, span, stride, has_seen = (int(x) for x in head[:6])
        keys = np.frombuffer(f.read(8 * n * nwk), dtype="<u8").reshape(n, nwk)
        w = np.frombuffer(f.read(4 * n), dtype="<u4")
        seen = np.frombuffer(f.read(16 * n), dtype="<u8").reshape(n, 2)
    return dict(n=n, nwk=nwk, k=k, span=span, stride=stride, keys=keys, w=w, seen=seen)


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    tmp = os.environ.get("TMPDIR", "/tmp")
    from diff_pair_store_records import __file__ as child_script
    for name in ("d0", "d2"):
        env = dict(os.environ, KATOME_DIST_ROUTE="tiles", KATOME_LIB=os.path.join(ROOT, "build_variants", "libkatome_gpu_%s.so" % name),
                   KATOME_DUMP_RECORDS=os.path.join(tmp, "rec_" + name))
        for old in glob.glob(os.path.join(tmp, "rec_%s.*.bin" % name)):
            os.remove(old)
        subprocess.run([sys.executable, child_script, "--child", str(reads)], env=env, timeout=900)
    right = {}
    for path in glob.glob(os.path.join(tmp, "rec_d0.*.bin")):
        d = read_dump(path)
        if d["stride"] != 1:
            continue
        for key, (a, b) in zip(d["keys"][:, 0], d["seen"]):
            right[(int(a), int(b))] = int(key)
    print("right records:", len(right))
    for path in sorted(glob.glob(os.path.join(tmp, "rec_d2.*.bin"))):
        d = read_dump(path)
        if d["stride"] != 1:
            continue
        span = d["span"]
        n_wrong = 0
        rows = []
        for i in range(d["n"]):
            a, b = int(d["seen"][i, 0]), int(d["seen"][i, 1])
            want = right.get((a, b))
            got = int(d["keys"][i, 0])
            if want is None or want == got:
                continue
            n_wrong += 1
            w = int(d["w"][i])
            p, block = w & 0xFFFF, w >> 16
            o = p % span
            sh = [t for t in range(0, 64) if (want >> t) == got]
            total = (sh[0] + 2 * (span - 1 - o)) if sh else -1
            rows.append((i, block, p, p >> 8, p & 255, p & 63, o, total, got, want))
        print(os.path.basename(path), "records", d["n"], "wrong", n_wrong)
        trips = sorted(set(r[3] for r in rows)); waves = sorted(set((r[4] >> 6) for r in rows))
        print("   loop trips (p >> 8) of the wrong records:", trips[:40], " waves of the block:", waves)
        agree_tid = sum(1 for r in rows if r[7] >= 0 and r[7] == r[5]); agree_p = sum(1 for r in rows if r[7] >= 0 and r[7] == r[2])
        known = sum(1 for r in rows if r[7] >= 0)
        print("   of %d wrong records with a recognisable shift: total shift == lane %d, total shift == p %d" % (known, agree_tid, agree_p))
        blocks = {}
        for r in rows:
            blocks.setdefault((r[1], r[3], r[4] >> 6), 0)
            blocks[(r[1], r[3], r[4] >> 6)] += 1
        print("   (block, trip, wave) groups:", len(blocks), " sizes:", sorted(blocks.values())[-10:])
        for r in rows[:80]:
            print("      i=%d block=%d p=%d trip=%d tid=%d lane=%d o=%d total_shift=%d got=%016x want=%016x" % r)


if __name__ == "__main__":
    main()
