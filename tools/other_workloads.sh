#!/bin/bash
# Runs on the GPU box (via gpurun): bench.py on the shapes beside C3 -- k = 40 (the reference's config.txt), 101-bp reads, k = 63 -- by packed
# key and in the reference's numbering, and an eighth of C3 through both sharded routes; one JSON line per run into
# gpurun_out/other_workloads.jsonl ({"run", "ms_per_step", "phases_ms", "counts"}); copy it to profiles/<round>_other_workloads.jsonl
set -o pipefail
OUT=gpurun_out/other_workloads.jsonl
mkdir -p gpurun_out; : > $OUT
run() {   # label, then bench.py arguments (and NAME=value environment in front of them)
  local label="$1"; shift
  local envs=()
  while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 "$@" > gpurun_out/_ow.json 2> gpurun_out/_ow.err || { echo "FAILED: $label"; tail -3 gpurun_out/_ow.err; return; }
  python - "$label" >> $OUT <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_ow.json").read().strip().splitlines()[-1])
print(json.dumps({"run": sys.argv[1], "ms_per_step": round(d["ms_per_step"], 1),
                  "phases_ms": {k: round(v["ms_per_step"], 1) for k, v in d["kernels"].items() if not k.startswith("k:")},
                  "counts": d.get("counts"), "exchange_ms": d.get("exchange_ms")}))
PY
  tail -1 $OUT | cut -c1-260
}
run "k40 50M reads, packed key" --workload k40 --reads 50000000
run "k40 50M reads, first-seen" --workload k40 --reads 50000000 --first-seen-order
run "r101 50M reads, packed key" --workload r101 --reads 50000000
run "r101 50M reads, first-seen" --workload r101 --reads 50000000 --first-seen-order
run "c5 25M reads (k=63), packed key" --workload c5 --reads 25000000
run "c5 25M reads (k=63), first-seen" --workload c5 --reads 25000000 --first-seen-order
run "C3 first-seen" --first-seen-order
run "C3 first-seen, tile levels in tables" KATOME_SORTED_TILES=0 --first-seen-order
run "C3 packed key, tile levels in tables" KATOME_SORTED_TILES=0
run "8-rank share of C3 (25M reads), level-by-level route, RCCL to itself" KATOME_DIST_ROUTE=tiles --force-dist --reads 25000000
run "same, one-exchange route" KATOME_DIST_ROUTE=local --force-dist --reads 25000000
run "same reads, one-GPU build" --reads 25000000
