"""Registers, scratch and spills of the library's kernels whose (demangled) name holds a pattern:
    python tools/kernel_regs.py PATTERN [libkatome_gpu.so]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pat = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "katome_amd", "lib", "libkatome_gpu.so")
data = open(lib, "rb").read()
import struct
at = 0
while True:
    at = data.find(MAGIC, at)
    if at < 0:
        break
    n = struct.unpack_from("<Q", data, at + 24)[0]
    p = at + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p); p += 24
        triple = data[p:p + tl].decode(); p += tl
        if "gfx950" not in triple:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(data[at + off: at + off + size]); path = f.name
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
        os.unlink(path)
        for b in notes.split(".agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", b)
            if not name:
                continue
            d = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
            if pat not in d:
                continue
            g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, b) or [None, "?"])[1]
            print("%-90s vgpr %s sgpr %s scratch %s vgpr_spill %s sgpr_spill %s lds %s" % (d.split("(")[0][:90], g("vgpr_count"), g("sgpr_count"),
                  g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size")))
    at += len(MAGIC)
