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


if __name__ == "__main__":
    sys.exit(main())
