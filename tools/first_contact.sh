#!/bin/bash
# First contact with a node that has more than one MI355X (nothing in this repository has run on two devices yet: DESIGN.md
# section 6).  Run on such a node from the repository root; every step appends to gpurun_out/first_contact/ and the summary is
# printed at the end.  Steps, each independent of the others' success:
#   1. RCCL between two DIFFERENT devices, messages below and above the 1 GiB cut (tools/probe_two_devices.py);
#   2. the sharded build with thread ranks on real devices against the one-GPU build (tools/check_sharded_scale.py, peer copies);
#   3. bench.py --gpus 2 / 4 / 8 (one process per GPU over RCCL; the line carries config.comm_ranks = the communicator's own
#      rank count and config.route), and the one-GPU line beside them.
set -u
OUT=gpurun_out/first_contact
mkdir -p $OUT
N=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "GPUs visible: $N" | tee $OUT/summary.txt
if [ "$N" -lt 2 ]; then echo "one GPU only: nothing to do" | tee -a $OUT/summary.txt; exit 0; fi
echo "== 1. RCCL between devices 0 and 1" | tee -a $OUT/summary.txt
timeout -k 10 600 python3 tools/probe_two_devices.py 0.25 1.5 3.0 > $OUT/two_devices.txt 2>&1; echo "rc=$?" | tee -a $OUT/summary.txt
grep "mismatching" $OUT/two_devices.txt | tee -a $OUT/summary.txt
echo "== 2. thread ranks on real devices (peer copies) against one GPU" | tee -a $OUT/summary.txt
W=$((N < 8 ? N : 8))
KATOME_REAL_DEVICES=1 timeout -k 10 900 python3 tools/check_sharded_scale.py 8000000 $W supermers,tiles > $OUT/sharded_scale.txt 2>&1; echo "rc=$?" | tee -a $OUT/summary.txt
grep -E "ranks|one GPU|OK|MISMATCH" $OUT/sharded_scale.txt | tee -a $OUT/summary.txt
echo "== 3. bench.py" | tee -a $OUT/summary.txt
for G in 1 2 4 8; do
  [ "$G" -gt "$N" ] && break
  timeout -k 10 900 python3 bench.py --gpus $G --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_$G.json 2> $OUT/bench_$G.err; RC=$?
  python3 - $OUT/bench_$G.json $G $RC <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("gpus %s: %.1f ms/step, %.3e k-mers/s, edges %d, transport %s, route %s, comm ranks %s" % (
        sys.argv[2], d["ms_per_step"], d["value"], d["distinct_edges"], d["config"].get("transport"), d["config"].get("route"), d["config"].get("comm_ranks")))
except Exception as e:
    print("gpus %s: rc=%s, no line (%s)" % (sys.argv[2], sys.argv[3], e))
PY
done
echo "summary in $OUT/summary.txt"
