"""Condenses the rocprofv3 output of tools/profile_round.sh into small files meant for profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]


def find(pattern):
    hits = glob.glob(os.path.join(out_dir, pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("katome::", "")
    return name.split("(")[0][:90]


lines = ["# rocprofv3 summary %s" % tag, ""]
stats = find("trace/**/*kernel_stats.csv")
if stats:
    lines += ["## kernel stats (rocprofv3 --kernel-trace --stats; whole bench.py run incl. its set-up and warm-up builds)", "",
              "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for row in csv.DictReader(open(stats)):
        lines.append("| %s | %s | %.2f | %.1f | %s |" % (short(row["Name"]), row["Calls"], float(row["TotalDurationNs"]) / 1e6,
                                                      float(row["AverageNs"]) / 1e3, row["Percentage"]))
    lines.append("")
trace = find("trace/**/*kernel_trace.csv")
if trace:
    per = defaultdict(list)
    for row in csv.DictReader(open(trace)):
        per[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), row.get("VGPR_Count", ""),
                                               row.get("LDS_Block_Size", ""), row.get("Grid_Size", "")))
    lines += ["## per-kernel durations from the dispatch trace", "", "| kernel | dispatches | avg us | min us | max us | VGPR | LDS |",
              "|---|---|---|---|---|---|---|"]
    for k, v in sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        d = [x[0] for x in v]
        lines.append("| %s | %d | %.1f | %.1f | %.1f | %s | %s |" % (k, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, v[0][1], v[0][2]))
    lines.append("")
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = find("pmc_%s/**/*counter_collection.csv" % c)
    if not f:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != c:
            continue
        a = acc[short(row["Kernel_Name"])]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
    pmc[c] = acc
if pmc:
    lines += ["## HBM traffic counters (separate --pmc passes; same workload, one step)", "",
              "FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; per the MI355X guide FETCH_SIZE reads half of the bytes",
              "of a wide coalesced stream on gfx950 (doubled in the 'corrected' column for streaming kernels only).", "",
              "| kernel | dispatches | FETCH_SIZE KiB/dispatch | WRITE_SIZE KiB/dispatch |", "|---|---|---|---|"]
    names = sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {})))
    summary = {}
    for k in names:
        f = pmc.get("FETCH_SIZE", {}).get(k, [0, 0])
        w = pmc.get("WRITE_SIZE", {}).get(k, [0, 0])
        n = max(f[1], w[1], 1)
        summary[k] = {"dispatches": n, "fetch_kib": f[0] / max(f[1], 1), "write_kib": w[0] / max(w[1], 1)}
        lines.append("| %s | %d | %.0f | %.0f |" % (k, n, summary[k]["fetch_kib"], summary[k]["write_kib"]))
    cfg, src = {}, None
    try:
        line = json.load(open(os.path.join(out_dir, "bench_under_pmc_FETCH_SIZE.json")))
        cfg, src = line["config"], line.get("source_id")
    except Exception:
        pass
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of "
                         "`bench.py --steps 1 --warmup 0`; per-dispatch averages, KiB; profiles/%s_summary.md" % tag,
               "config": {k: cfg.get(k) for k in ("reads", "read_len", "k", "batch_reads", "tile_span")},
               "source_id": src,        # bench.py source_id(): the kernels these counters were taken on
               "kernels": summary}, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)
    lines.append("")
for f in ("bench_under_trace.json",):
    p = os.path.join(out_dir, f)
    if os.path.exists(p) and os.path.getsize(p):
        lines += ["## bench.py line of the traced run", "", "```", open(p).read().strip(), "```", ""]
open(os.path.join(out_dir, "summary_%s.md" % tag), "w").write("\n".join(lines))
print("\n".join(lines[:60]))
