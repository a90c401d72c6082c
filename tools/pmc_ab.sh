#!/bin/bash
# usage: pmc_ab.sh <tag> [KATOME_LIB]
set -o pipefail
TAG=$1; LIB=$2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3k/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export KATOME_LIB=$REPO/$LIB
python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/pmc_$C -o pmc -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err || { tail -3 $OUT/pmc_$C.err; }
done
