#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; never combined with API tracing).
# Usage: tools/profile_round.sh <round-tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="${@:---steps 2 --warmup 1 --no-cpu-baseline --no-extras}"
echo "[profile] kernel trace: bench.py $BENCH_ARGS"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $BENCH_ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
# PMC passes: the same workload, one step (counter collection serialises dispatches)
PMC_ARGS="${PMC_ARGS:---steps 1 --warmup 0 --no-cpu-baseline --no-extras}"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[profile] pmc $C: bench.py $PMC_ARGS"
  rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/pmc_$C -o pmc -- python3 $REPO/bench.py $PMC_ARGS > $OUT/bench_under_pmc_$C.json 2> $OUT/pmc_$C.err || { tail -5 $OUT/pmc_$C.err; exit 1; }
done
cd $REPO && python3 tools/summarize_profile.py $OUT $TAG
