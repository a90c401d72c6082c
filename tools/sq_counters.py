"""Per-dispatch averages of the SQ counters tools/sq_counters.sh collected, for the kernels the build spends its time in:
    python tools/sq_counters.py gpurun_out/sq_<tag>"""
import csv, glob, os, sys
KERNELS = ["lds_count_packed_kernel<true>", "lds_count_full_kernel<false, false>", "lds_count_wide_kernel<false, 13, 2, false>", "lds_count_kernel<true, 13, false>",
           "radix_scatter_kernel<1, true, katome::RadixDigit<1>, true>", "radix_scatter_kernel<1, true, katome::HashDigit<1>, true>", "dst_merge_kernel<1, false>",
           "run_sort_wave_kernel<1, true>"]
csv.field_size_limit(1 << 30)
acc = {}          # kernel -> counter -> [sum, n]
dur = {}          # kernel -> [sum ns, n]
for path in sorted(glob.glob(os.path.join(sys.argv[1], "p*", "**", "pmc_counter_collection.csv"), recursive=True)):
    seen = set()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        k = next((x for x in KERNELS if x in name), None)
        if not k:
            continue
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
        key = (path, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            d = dur.setdefault(k, [0.0, 0])
            d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
counters = sorted({c for k in acc.values() for c in k})
ks = [k for k in KERNELS if k in acc]
print("| counter | " + " | ".join("`%s` (%.1f ms under the counters)" % (k.replace("katome::", ""), dur[k][0] / dur[k][1] / 1e6) for k in ks) + " |")
print("|---|" + "---|" * len(ks))
for c in counters:
    print("| %s | " % c + " | ".join(("%.3g" % (acc[k][c][0] / acc[k][c][1])) if c in acc[k] else "" for k in ks) + " |")
