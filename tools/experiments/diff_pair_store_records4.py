"""Fourth look (see diff_pair_store_records.py): build d4 is d1's own assembly with one instruction added -- the weight store
carries HW_REG_GPR_ALLOC / HW_ID / XCC_ID of the wave instead of the weight (registers and code otherwise unchanged: 32 VGPRs, the
shift amount in v31).  Where do the waves that write wrong keys run, and where do their registers sit?"""
import collections, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from diff_pair_store_records3 import read_dump


def fields(w):
    return dict(vgpr_base=w & 63, vgpr_size=(w >> 6) & 63, wave=(w >> 12) & 15, simd=(w >> 16) & 3, pipe=(w >> 18) & 3, cu=(w >> 20) & 15,
                sh=(w >> 24) & 1, se=(w >> 25) & 7, xcc=(w >> 28) & 15)


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    tmp = os.environ.get("TMPDIR", "/tmp")
    from diff_pair_store_records import __file__ as child_script
    for name in ("d0", "d4"):
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
    for path in sorted(glob.glob(os.path.join(tmp, "rec_d4.*.bin"))):
        d = read_dump(path)
        if d["stride"] != 1:
            continue
        w = d["w"].astype(np.uint32)
        print(os.path.basename(path), "records", d["n"])
        print("   all records: vgpr_base histogram", sorted(collections.Counter((w & 63).tolist()).items()))
        print("   all records: vgpr_size histogram", sorted(collections.Counter(((w >> 6) & 63).tolist()).items()))
        print("   all records: wave-slot histogram", sorted(collections.Counter(((w >> 12) & 15).tolist()).items()))
        wrong = []
        for i in range(d["n"]):
            want = right.get((int(d["seen"][i, 0]), int(d["seen"][i, 1])))
            if want is not None and want != int(d["keys"][i, 0]):
                wrong.append(i)
        print("   wrong records:", len(wrong))
        groups = collections.Counter(int(w[i]) for i in wrong)
        for info, cnt in groups.most_common(40):
            print("      %3d wrong records from a wave with" % cnt, fields(info))


if __name__ == "__main__":
    main()
