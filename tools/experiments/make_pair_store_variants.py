"""Builds diagnostic variants of the library that differ ONLY in how expand_tiles_kernel<.., TO_TABLE=false> writes a record's
two sequence numbers (table.hip; round 2's "one tile in 5000 lost" finding), into build_variants/ (git-ignored, travels with
gpurun):

  libkatome_gpu_v1.so   pair read from LDS AFTER the key/weight stores, ONE 16-byte plain store   (the variant that lost tiles)
  libkatome_gpu_v2.so   pair read AFTER the stores, two 8-byte plain stores
  libkatome_gpu_v3.so   pair read BEFORE the stores, one 16-byte plain store
  libkatome_gpu_v4.so   pair read BEFORE the stores, two 8-byte plain stores (shipped: two 8-byte agent-scope stores)
  libkatome_gpu_v5.so   as v1, with `s_waitcnt vmcnt(0)` between the key/weight stores and the LDS read of the pair
  libkatome_gpu_v7.so   as v1, with a compiler barrier only (asm volatile("" ::: "memory")) in that place: no hardware wait
  libkatome_gpu_v8.so   as v1, with `s_waitcnt vmcnt(1)` there: the key store has completed, the weight store may be in flight

  libkatome_gpu_d0.so / _d1.so   the shipped kernel / v1, both with -DKATOME_DEBUG_DUMP: KATOME_DUMP_RECORDS=<prefix> writes the
                        records of every expansion to <prefix>.<call>.bin (tools/diff_pair_store_records.py compares them)

Run a first-seen-order sharded build against each with KATOME_LIB=build_variants/libkatome_gpu_vN.so (tools/check_pair_store.py).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "katome_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")

EARLY = ("if (!TO_TABLE && tile_seen) { seq_fwd = lseen[2 * t] + (u64)o * stride; "
         "seq_rev = lseen[2 * t + 1] + (u64)(span - 1 - o) * stride; }")
LATE = ("seq_fwd = lseen[2 * t] + (u64)o * stride; seq_rev = lseen[2 * t + 1] + (u64)(span - 1 - o) * stride;")
STORE16 = ("ulonglong2 v; v.x = flipped ? seq_rev : seq_fwd; v.y = flipped ? seq_fwd : seq_rev; "
           "*reinterpret_cast<ulonglong2*>(&kmer_seen[2 * (bbase + p)]) = v;")
STORE8 = ("kmer_seen[2 * (bbase + p)] = flipped ? seq_rev : seq_fwd; kmer_seen[2 * (bbase + p) + 1] = flipped ? seq_fwd : seq_rev;")


WAITS = {False: "", True: 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ', "barrier": 'asm volatile("" ::: "memory"); ',
         "vm1": 'asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); '}


def variant(src, late, wide, wait=False):
    a = src.index("                if (tile_seen) {        // (kmer_seen is the records'")
    b = src.index("        __syncthreads();\n    }\n    if (TO_TABLE) {")
    body = "                if (tile_seen) {\n                    %s\n                    %s\n                }\n            }\n        }\n" % (
        WAITS[wait] + (LATE if late else ""), STORE16 if wide else STORE8)
    out = src[:a] + body + src[b:]
    if late:
        assert EARLY in out
        out = out.replace(EARLY, "")
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "table.hip")).read()
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "table.o"]
    for name, late, wide, wait in (("v1", True, True, False), ("v2", True, False, False), ("v3", False, True, False), ("v4", False, False, False),
                                   ("v5", True, True, True), ("v7", True, True, "barrier"), ("v8", True, True, "vm1")):
        hip = os.path.join(OUT, "table_%s.hip" % name)
        open(hip, "w").write(variant(src, late, wide, wait))
        obj = os.path.join(OUT, "table_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
                               "-I", CSRC, "-c", hip, "-o", obj])
        so = os.path.join(OUT, "libkatome_gpu_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs + ["-ldl", "-lpthread"])
        print("built", so)


def top_register_builds():
    """v9: v1 with a register beyond its 32 named at the kernel's start (the allocation becomes 40: v31 is no longer the top one);
    v10: v1 with v31 declared clobbered inside the loop (the shift amount has to live elsewhere)"""
    src = open(os.path.join(CSRC, "table.hip")).read()
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "table.o"]
    v1 = variant(src, True, True, False)
    head = "    __shared__ u64 lkey[BLOCK * NWT];\n    __shared__ u64 lseen[BLOCK * 2];      // first-seen-order mode"
    assert head in v1
    v9 = v1.replace(head, '    if (!TO_TABLE) asm volatile("v_mov_b32 v39, 0" ::: "v39");\n' + head, 1)
    loop = "            Key<NWK> x = sub_window<NWT, NWK>(tk, k, span, stride, o);"
    assert loop in v1
    v10 = v1.replace(loop, '            if (!TO_TABLE) asm volatile("" ::: "v31");\n' + loop, 1)
    for name, text in (("v9", v9), ("v10", v10)):
        hip = os.path.join(OUT, "table_%s.hip" % name)
        open(hip, "w").write(text)
        obj = os.path.join(OUT, "table_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
                               "-I", CSRC, "-c", hip, "-o", obj])
        so = os.path.join(OUT, "libkatome_gpu_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs + ["-ldl", "-lpthread"])
        print("built", so)


def dump_builds():
    src = open(os.path.join(CSRC, "table.hip")).read()
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "table.o"]
    for name, text in (("d0", src), ("d1", variant(src, True, True, False))):
        hip = os.path.join(OUT, "table_%s.hip" % name)
        open(hip, "w").write(text)
        obj = os.path.join(OUT, "table_%s.o" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
                               "-DKATOME_DEBUG_DUMP", "-I", CSRC, "-c", hip, "-o", obj])
        so = os.path.join(OUT, "libkatome_gpu_%s.so" % name)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs + ["-ldl", "-lpthread"])
        print("built", so)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "top":
        os.makedirs(OUT, exist_ok=True)
        top_register_builds()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dump":
        os.makedirs(OUT, exist_ok=True)
        dump_builds()
        sys.exit(0)
    sys.exit(main())
