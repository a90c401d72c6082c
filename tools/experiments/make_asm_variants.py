"""Variants of the bad kernel shape (tools/make_pair_store_variants.py, v1/d1) made by editing its ASSEMBLY, so that register
allocation and instruction order stay exactly as in the failing build except for the one thing under test.  The compile is
replayed from hipcc's own -save-temps steps (assemble, link, bundle, host compile).  Output: build_variants/libkatome_gpu_a<X>.so

  aA  only the kernel descriptor changes: 40 registers allocated instead of 32 (not an instruction differs)
  aC  s_nop 7 after the v_mul_lo_u32 that writes v31, before its first reader
  aD  v_lshlrev_b32 v31, 1, v29 instead of v_mul_lo_u32 v31, v29, s33 (stride 1: the same value from a full-rate operation)
  aF  two s_nop 7 behind the second barrier of the outer loop
  aG  v_mov_b32 v31, 37 in the inner loop's pre-header (behind the barrier)
  aH  the shift amount lives in v28 and the tile index in v31 (the two registers exchanged throughout the kernel)
  aI  s_waitcnt lgkmcnt(0) before the v_mul_lo_u32 that writes v31 (no LDS read in flight when v31 is written and read)
"""
import os, re, shlex, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "katome_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants")
WORK = "/tmp/asm_variants"
NAME = "_ZN6katome19expand_tiles_kernelILi2ELi1ELb1ELb0EEEvPKNS_6SlotOfIXT_EE4typeEmmjjjPNS1_IXT0_EE4typeEmPmPjS9_SA_S9_PKmS9_"
ASM = "table_d1-hip-amdgcn-amd-amdhsa-gfx950.s"


def sh(cmd, **kw):
    return subprocess.run(cmd, shell=True, cwd=WORK, check=True, capture_output=True, text=True, **kw)


def prepare():
    os.makedirs(WORK, exist_ok=True)
    src = os.path.join(OUT, "table_d1.hip")
    if not os.path.exists(src):
        raise SystemExit("run tools/make_pair_store_variants.py dump first")
    sh("cp %s table_d1.hip" % shlex.quote(src))
    r = subprocess.run("/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I %s -c table_d1.hip -o table_d1.o --save-temps -v" % CSRC,
                       shell=True, cwd=WORK, capture_output=True, text=True)
    lines = [l for l in r.stderr.splitlines() if l.startswith(' "')]
    steps = {}
    for l in lines:
        if "-cc1as" in l and "amdgcn" in l: steps["as"] = l
        elif "lld" in l.split()[0]: steps["ld"] = l
        elif "clang-offload-bundler" in l: steps["bundle"] = l
        elif "-fcuda-include-gpubinary" in l and "-emit-llvm-bc" in l: steps["host_bc"] = l
        elif "-triple x86_64" in l and " -S " in l and "-cc1 " in l: steps["host_s"] = l
        elif "-cc1as" in l and "x86_64" in l: steps["host_as"] = l
    assert len(steps) == 6, sorted(steps)
    sh("cp %s orig.s" % ASM)
    return steps


def kernel_span(s):
    a = s.index("\n" + NAME + ":")
    return a, s.index(".Lfunc_end", a)


def patch(s, which):
    a, b = kernel_span(s)
    body = s[a:b]
    def once(old, new):
        nonlocal body
        assert body.count(old) == 1, (which, old, body.count(old))
        body = body.replace(old, new)
    if which == "A":
        pass
    elif which == "C":
        once("\tv_mul_lo_u32 v31, v29, s33\n", "\tv_mul_lo_u32 v31, v29, s33\n\ts_nop 7\n")
    elif which == "D":
        once("\tv_mul_lo_u32 v31, v29, s33\n", "\tv_lshlrev_b32_e32 v31, 1, v29\n")
    elif which == "F":
        once("\ts_barrier\n\ts_and_saveexec_b64 s[40:41], vcc\n", "\ts_barrier\n\ts_nop 7\n\ts_nop 7\n\ts_and_saveexec_b64 s[40:41], vcc\n")
    elif which == "G":
        once("\tv_mov_b32_e32 v27, v26\n", "\tv_mov_b32_e32 v27, v26\n\tv_mov_b32_e32 v31, 37\n")
    elif which == "H":
        body = re.sub(r"\bv31\b", "vTMP", body); body = re.sub(r"\bv28\b", "v31", body); body = body.replace("vTMP", "v28")
    elif which == "I":
        once("\tv_mul_lo_u32 v31, v29, s33\n", "\ts_waitcnt lgkmcnt(0)\n\tv_mul_lo_u32 v31, v29, s33\n")
    else:
        raise SystemExit("unknown variant " + which)
    s = s[:a] + body + s[b:]
    if which == "A":
        a2 = s.index(".amdhsa_kernel " + NAME); b2 = s.index(".end_amdhsa_kernel", a2)
        kd = s[a2:b2]
        assert ".amdhsa_next_free_vgpr 32" in kd and ".amdhsa_accum_offset 32" in kd
        s = s[:a2] + kd.replace(".amdhsa_next_free_vgpr 32", ".amdhsa_next_free_vgpr 40").replace(".amdhsa_accum_offset 32", ".amdhsa_accum_offset 40") + s[b2:]
    return s


def main():
    steps = prepare()
    orig = open(os.path.join(WORK, "orig.s")).read()
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f != "table.o"]
    for which in (sys.argv[1:] or list("ACDFGHI")):
        open(os.path.join(WORK, ASM), "w").write(patch(orig, which))
        for st in ("as", "ld", "bundle", "host_bc", "host_s", "host_as"):
            sh(steps[st])
        so = os.path.join(OUT, "libkatome_gpu_a%s.so" % which)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, os.path.join(WORK, "table_d1.o")] + objs + ["-ldl", "-lpthread"])
        print("built", so, flush=True)


if __name__ == "__main__":
    main()
