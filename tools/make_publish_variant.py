"""Builds build_variants/libkatome_gpu_ra.so: the library with the 128-/192-bit slot publication of table.hip written with
C++-memory-model release/acquire atomics (agent scope) instead of `s_waitcnt vmcnt(0)` + relaxed agent-scope stores and plain
cached reads -- for an A/B of the insert phases (`KATOME_LIB=build_variants/libkatome_gpu_ra.so python bench.py ...`)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "katome_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")


def main():
    os.makedirs(OUT, exist_ok=True)
    s = open(os.path.join(CSRC, "table.hip")).read()
    pub = '''                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_agent(&slots[s].hi, key.w[0] | OCC);'''
    assert s.count(pub) == 2
    s = s.replace(pub, "                __hip_atomic_store(&slots[s].hi, key.w[0] | OCC, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);")
    rd = '''        u64 cur = slots[s].hi;
        bool cached_view = true;'''
    assert s.count(rd) == 2
    s = s.replace(rd, '''        u64 cur = __hip_atomic_load(&slots[s].hi, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        bool cached_view = true;''')
    hip = os.path.join(OUT, "table_ra.hip")
    open(hip, "w").write(s)
    obj = os.path.join(OUT, "table_ra.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function",
                           "-I", CSRC, "-c", hip, "-o", obj])
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "table.o"]
    so = os.path.join(OUT, "libkatome_gpu_ra.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs + ["-ldl", "-lpthread"])
    print("built", so)


if __name__ == "__main__":
    main()
