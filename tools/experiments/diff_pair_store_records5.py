"""Fifth look (see diff_pair_store_records4.py): build d5 = d4 + `v_mov_b32 v31, 37` in the kernel's prologue.  If the wrong keys
are now the tile shifted by 37 in every lane, the wave's FIRST write to v31 (v_mul_lo_u32) does not land and the register keeps what
it held; if they are still shifted by the lane id, something writes the lane id into v31."""
import collections, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from diff_pair_store_records3 import read_dump


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    lib = sys.argv[2] if len(sys.argv) > 2 else "d5"
    tmp = os.environ.get("TMPDIR", "/tmp")
    from diff_pair_store_records import __file__ as child_script
    for name in ("d0", lib):
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
    hist = collections.Counter()
    n_wrong = 0
    for path in sorted(glob.glob(os.path.join(tmp, "rec_%s.*.bin" % lib))):
        d = read_dump(path)
        if d["stride"] != 1:
            continue
        span = d["span"]
        for i in range(d["n"]):
            want = right.get((int(d["seen"][i, 0]), int(d["seen"][i, 1])))
            got = int(d["keys"][i, 0])
            if want is None or want == got:
                continue
            n_wrong += 1
            sh = [t for t in range(0, 64) if (want >> t) == got]
            hist[sh[0] if sh else -1] += 1
    print(lib, "wrong records", n_wrong, "; extra shift of the wrong key against the right one (-1: not a plain shift, e.g. the other strand):")
    print("  ", sorted(hist.items()))


if __name__ == "__main__":
    main()
