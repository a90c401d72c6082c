"""Scans the gfx950 ISA of the library's kernels for the code shape round 3 tied the lost records of expand_tiles_kernel to
(table.hip): an LDS load whose destination registers are the address or data registers of a global store issued earlier in the
same straight-line stretch, with no `s_waitcnt vmcnt(0)` in between.  Linear scan per function (labels reset the state: a
conservative view of control flow).  Usage: hipcc -S --cuda-device-only each .hip into a directory, then this script on it."""
import glob
import os
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    hits, func, pending = [], None, []
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if re.match(r"^_Z[\w.$]+:$", t):
            func, pending = t[:-1], []
            continue
        if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("s_barrier"):
            pending = []                                  # (a branch target or a barrier: start over)
            continue
        op = t.split()[0] if t else ""
        if op.startswith("global_store") or op.startswith("global_atomic"):
            args = t[len(op):].split(",")
            used = set()
            for a in (args[:1] if ADDRESS_ONLY else args[:2]):      # global_store vaddr, vdata, ...
                used |= regs(a)
            pending.append((n, t, used))
        elif op.startswith("s_waitcnt") and "vmcnt(0)" in t:
            pending = []
        elif op.startswith("ds_read") or (op.startswith("ds_") and "_rtn" in op):
            dst = regs(t[len(op):].split(",")[0])
            for sn, st, used in pending:
                if dst & used:
                    hits.append((func, n, t, sn, st))
    return hits


ADDRESS_ONLY = "--address-only" in sys.argv

if __name__ == "__main__":
    total = 0
    for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.s"))):
        for func, n, t, sn, st in scan(f):
            total += 1
            print("%s: %s\n    line %d: %s\n    after line %d: %s" % (os.path.basename(f), (func or "?")[:90], n, t, sn, st))
    print("%d LDS loads into the %sregisters of a global store still in flight" % (total, "address " if ADDRESS_ONLY else ""))
