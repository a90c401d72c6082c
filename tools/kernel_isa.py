"""Disassembly of the library's kernels whose (mangled) name holds a pattern:  python tools/kernel_isa.py PATTERN [lib.so] > out.s"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pat = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "katome_amd", "lib", "libkatome_gpu.so")
data = open(lib, "rb").read()
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
        syms = subprocess.run(["nm", path], capture_output=True, text=True).stdout
        for line in syms.splitlines():
            parts = line.split()
            if len(parts) == 3 and parts[1] in "Tt" and pat in parts[2] and not parts[2].endswith(".kd"):
                out = subprocess.run([LLVM + "/llvm-objdump", "-d", "--disassemble-symbols=" + parts[2], path], capture_output=True, text=True).stdout
                print(out)
        os.unlink(path)
    at += len(MAGIC)
