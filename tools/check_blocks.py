import sys, re
live = {}
bad = 0
for ln, line in enumerate(open(sys.argv[1])):
    m = re.match(r"\[katome block\] \+ (0x[0-9a-f]+) (\d+) dev (\d+) stream (\S+)", line)
    if m:
        p, n = int(m.group(1), 16), int(m.group(2))
        for q, (m2, l2) in live.items():
            if p < q + m2 and q < p + n:
                print("OVERLAP line", ln, hex(p), n, "with", hex(q), m2, "from line", l2); bad += 1
        live[p] = (n, ln)
        continue
    m = re.match(r"\[katome block\] - (0x[0-9a-f]+)", line)
    if m:
        p = int(m.group(1), 16)
        if p not in live:
            print("free of unknown", hex(p), "line", ln)
        live.pop(p, None)
print("blocks checked, overlaps:", bad, "live at end:", len(live))
