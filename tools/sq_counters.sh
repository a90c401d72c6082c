#!/bin/bash
# SQ counters of the build's kernels (rocprofv3 --pmc, three counters a pass, never with API tracing): bash tools/sq_counters.sh <tag>
# -> gpurun_out/sq_<tag>/table.md (per-dispatch averages of the kernels tools/sq_counters.py lists)
set -o pipefail
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  P=$((P + 1))
  echo "[sq] pass $P: $C"
  rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/p$P -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/bench_p$P.json 2> $OUT/p$P.err || { tail -5 $OUT/p$P.err; exit 1; }
done
cd $REPO && python3 tools/sq_counters.py $OUT > $OUT/table.md && cat $OUT/table.md
