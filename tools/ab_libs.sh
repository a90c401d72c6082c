#!/bin/bash
# A/B of library variants on the headline build: tools/ab_libs.sh OUTDIR name=path.so ...   ("main" = the shipped library)
set -o pipefail
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
    name=${v%%=*}; lib=${v#*=}
    if [ "$lib" = "main" ]; then unset KATOME_LIB; else export KATOME_LIB="$PWD/$lib"; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 ${AB_ARGS:-} > "$out/$name.json" 2> "$out/$name.err" || { echo "$name: bench failed"; tail -3 "$out/$name.err"; exit 1; }
    python - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
kl = {k: round(v["ms_per_step"], 2) for k, v in d["kernel_launches"].items() if "lds_count" in k}
print(sys.argv[2], round(d["ms_per_step"], 1), kl, json.dumps(d.get("lc_phases", {})))
PY
done
