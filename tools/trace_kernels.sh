#!/bin/bash
# rocprofv3 kernel trace of one bench.py step; prints the per-kernel averages (runs on the GPU box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/ktrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > $OUT/bench.json 2> $OUT/err.log || tail -5 $OUT/err.log
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for row in list(csv.DictReader(open(f)))[:40]:
    print("%-70s calls %4s avg %10.1f us total %9.2f ms" % (row["Name"].replace("katome::","").split("(")[0][:70], row["Calls"], float(row["AverageNs"])/1e3, float(row["TotalDurationNs"])/1e6))
PY
