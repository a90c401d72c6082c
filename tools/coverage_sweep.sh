#!/bin/bash
# Inputs that are not 300-fold coverage: C3's read count over longer genomes.
# Writes one bench line per genome length plus the card's peak memory use, sampled by rocm-smi at 5 Hz.
#   gpurun -- bash tools/coverage_sweep.sh [reads] [genome lengths...]
set -u
OUT=gpurun_out/coverage_sweep
mkdir -p $OUT
READS=${1:-200000000}; shift || true
LENS=${@:-"1000000000 2000000000"}
for G in $LENS; do
  ( while true; do rocm-smi --showmeminfo vram --json 2>/dev/null | python3 -c '
import sys, json
try:
    d = json.load(sys.stdin); c = next(iter(d.values()))
    print(int(c["VRAM Total Used Memory (B)"]))
except Exception: pass'; sleep 0.2; done ) > $OUT/mem_$G.txt &
  SAMPLER=$!
  KATOME_LEVEL_TRACE=1 timeout -k 10 420 python3 bench.py --reads $READS --genome-len $G --steps 2 --warmup 0 --no-cpu-baseline --no-extras \
      > $OUT/line_$G.json 2> $OUT/err_$G.log
  RC=$?
  kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
  PEAK=$(sort -n $OUT/mem_$G.txt | tail -1)
  echo "{\"reads\": $READS, \"genome_len\": $G, \"rc\": $RC, \"peak_vram_bytes\": ${PEAK:-null}}" >> $OUT/summary.jsonl
  echo "genome $G rc=$RC peak=${PEAK:-?}"; grep -h "katome levels" $OUT/err_$G.log | sort | uniq -c; tail -2 $OUT/err_$G.log; python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(json.dumps({k: d[k] for k in (\"ms_per_step\", \"distinct_edges\", \"nodes\", \"counts\")}))" $OUT/line_$G.json 2>/dev/null
  if [ $RC -ne 0 ] && [ $RC -ne 1 ]; then echo "stopping after rc=$RC"; break; fi
done
