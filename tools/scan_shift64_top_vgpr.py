"""Build-time guard against the gfx950 erratum that round 3 traced the lost records of expand_tiles_kernel to
(profiles/r03_shift64_erratum.md; reproducer: tools/probe_shift64_top_vgpr.hip): a 64-bit shift (v_lshlrev_b64, v_lshrrev_b64,
v_ashrrev_i64) whose AMOUNT is held in the last VGPR of the wave's allocation -- register 8n+7 with the next one not used by the
kernel -- sometimes shifts by v0 (the thread id) instead.  LLVM knows the erratum as Shift64HighRegBug and works around it on
gfx90a (GCNHazardRecognizer::fixShift64HighRegBug), not on gfx950, so the library checks its own ISA:

    python tools/scan_shift64_top_vgpr.py [libkatome_gpu.so | file.s ...]      (default: katome_amd/lib/libkatome_gpu.so)

A shared library (or object) is searched for its gfx950 code objects (clang offload bundles), which are disassembled with
llvm-objdump; a .s file is hipcc -S --cuda-device-only output.  Exit code 1 and one line per offending instruction if a kernel has
the shape.  The remedy for a flagged kernel is KATOME_SHIFT64_GUARD (common.h) at the top of the kernel: it names the next
register, which moves the end of the allocation a granule up and away from the amount."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
SHIFTS = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
STRICT = os.environ.get("KATOME_SCAN_STRICT") == "1"      # LLVM's rule: also a hit when no instruction names the next register


def is_hit(reg, arch_vgprs, used=None):
    """LLVM's condition for gfx90a (GCNHazardRecognizer::fixShift64HighRegBug): the amount is the last register of an allocation
    block (8n + 7) and the next ARCHITECTURAL register is not in use -- beyond the kernel's arch-VGPR count (the unified file's
    AGPRs behind it do not count), or, when the registers the function names are known, named by no instruction of it"""
    if (reg & 7) != 7:
        return False
    return reg + 1 >= arch_vgprs or (used is not None and (reg + 1) not in used)


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def scan_lines(lines, vgprs, label_re):
    """vgprs: function -> arch-VGPR count.  Two passes per function: the VGPRs any instruction names, then the shifts"""
    funcs, func = {}, None                    # function -> [(line number, text)]
    for n, line in enumerate(lines, 1):
        t = line.strip()
        m = label_re.match(t)
        if m:
            func = m.group(1)
            funcs.setdefault(func, [])
            continue
        if func is not None and t and not t.startswith((".", "//", ";")):
            funcs[func].append((n, t.split("//")[0].strip()))
    hits = []
    for func, body in funcs.items():
        if func not in vgprs:
            continue
        used = set()
        for _, t in body:
            for m in _VREG.finditer(t):
                if m.group(1) is not None:
                    used.add(int(m.group(1)))
                else:
                    used.update(range(int(m.group(2)), int(m.group(3)) + 1))
        for n, t in body:
            op = t.split()[0] if t else ""
            if op.replace("_e64", "") in SHIFTS:
                args = [a.strip() for a in t[len(op):].split(",")]
                m = re.fullmatch(r"v(\d+)", args[1]) if len(args) > 1 else None
                # (`used` is not consulted: the hardware's condition is the allocation's end, and KATOME_SHIFT64_GUARD moves that
                # end by clobbering the next register without an instruction that names it; pass it to apply LLVM's stricter rule)
                if m and is_hit(int(m.group(1)), vgprs[func], used if STRICT else None):
                    hits.append((func, n, t, vgprs[func]))
    return hits


_SAVEEXEC = re.compile(r"s_and_saveexec_b64 (s\[\d+:\d+\]), ")


def scan_empty_masked_regions(lines, label_re):
    """Second shape (profiles/r04_wrong_code.md): `s_and_saveexec_b64 sX, cond` followed at once by `s_or_b64 exec, exec, sX` -- a
    masked region with nothing in it.  hipcc 7.2 leaves that behind where it has DROPPED an assignment made under
    `uniform_flag || divergent_condition`: the value is set on the uniform-true edge only, the lanes of the divergent edge keep the
    default (lds_count_wide_kernel's read-out with `(k & 1) ||` in front of the palindrome test: every k-mer of an even k came out as
    its own reverse complement).  No kernel of the library has the shape; a new one fails the build and gets looked at."""
    hits, func = [], None
    body = [(n, line.strip()) for n, line in enumerate(lines, 1)]
    for idx, (n, t) in enumerate(body):
        m = label_re.match(t)
        if m:
            func = m.group(1)
            continue
        m = _SAVEEXEC.search(t)
        if not m or func is None:
            continue
        j = idx + 1
        while j < len(body) and (not body[j][1] or body[j][1].startswith((";", "//"))):
            j += 1
        if j < len(body) and re.search(r"s_or_b64 exec, exec, " + re.escape(m.group(1)) + r"\s*($|;|//)", body[j][1]):
            hits.append((func, n, t.split("//")[0].strip(), 0))
    return hits


def scan_asm(path):
    text = open(path).read()
    vgprs = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        total = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)).group(1))
        acc = re.search(r"\.amdhsa_accum_offset (\d+)", m.group(2))       # (where the AGPRs begin in the unified file, a multiple of 4)
        vgprs[m.group(1)] = min(total, int(acc.group(1))) if acc else total
    label = re.compile(r"^(_Z[\w.$]+):")
    return scan_lines(text.splitlines(), vgprs, label) + [h[:3] + (-1,) for h in scan_empty_masked_regions(text.splitlines(), label)], len(vgprs)


def code_objects(path):
    """the gfx950 code objects bundled into a host object or shared library"""
    data = open(path, "rb").read()
    out, i = [], data.find(MAGIC)
    while i >= 0:
        n = struct.unpack_from("<Q", data, i + 24)[0]
        q = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        i = data.find(MAGIC, i + 1)
    return out


def scan_binary(path):
    hits, kernels = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for k, co in enumerate(code_objects(path)):
            elf = os.path.join(tmp, "co%d.elf" % k)
            open(elf, "wb").write(co)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True, check=True).stdout
            vgprs, agprs, name, pending_agprs = {}, {}, None, 0
            for line in notes.splitlines():
                m = re.match(r"\s*-\s*\.agpr_count:\s+(\d+)", line) or re.match(r"\s*\.agpr_count:\s+(\d+)", line)
                if m:                                  # (a kernel's keys come sorted: .agpr_count ahead of its .name)
                    pending_agprs = int(m.group(1))
                m = re.match(r"\s*-?\s*\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                    agprs[name], pending_agprs = pending_agprs, 0
                m = re.match(r"\s*-?\s*\.vgpr_count:\s+(\d+)", line)
                if m and name:
                    vgprs[name] = int(m.group(1))
            for fn in vgprs:                      # (.vgpr_count is the unified total: the arch registers are what is left of it)
                vgprs[fn] -= agprs.get(fn, 0)
            kernels += len(vgprs)
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", elf], capture_output=True, text=True, check=True).stdout
            label = re.compile(r"^[0-9a-f]+ <([\w.$]+)>:")
            hits += scan_lines(dis.splitlines(), vgprs, label)
            hits += [h[:3] + (-1,) for h in scan_empty_masked_regions(dis.splitlines(), label)]
    return hits, kernels


def main(argv):
    files = argv[1:] or [os.path.join(ROOT, "katome_amd", "lib", "libkatome_gpu.so")]
    bad = kernels = 0
    for f in files:
        hits, n = scan_asm(f) if f.endswith(".s") else scan_binary(f)
        kernels += n
        for func, line, t, v in hits:
            bad += 1
            if v < 0:
                print("%s: %s\n    in %s (an empty masked region: a masked assignment was dropped -- profiles/r04_wrong_code.md)" % (os.path.basename(f), t, func[:110]))
            else:
                print("%s: %s\n    in %s (%d VGPRs in use: the amount is the last register of its allocation)" % (os.path.basename(f), t, func[:110], v))
    print("%d kernels scanned; %d 64-bit shifts with their amount in the last allocated VGPR or empty masked regions" % (kernels, bad))
    if kernels == 0:
        print("no gfx950 kernels found in", files)
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
