#!/bin/bash
# usage: tools/r3_tail_cus.sh <tag> "<cu counts>" ["<ilp values>"]  (GPU box) bench line against the CUs given to the tail stream
export TMPDIR=/tmp
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
for ilp in ${3:-1}; do
for n in ${2:-0 16 32}; do
  NIMBLE_DEDUP_ILP=$ilp NIMBLE_TAIL_CUS=$n timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --cpu-sample 0 --e2e-reads 0 > $OUT/bench_${ilp}_$n.log 2> $OUT/bench_${ilp}_$n.err
  python3 - $OUT/bench_${ilp}_$n.log $n $ilp <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ilp", sys.argv[3], "tail_cus", sys.argv[2], "ms_per_step %.3f" % j["ms_per_step"], "value %.3g" % j["value"], "stage", j.get("stage_ms"), "step min/med %.3f %.3f" % (j["step_ms"]["min"], j["step_ms"]["median"]))
except Exception as e:
    print("tail_cus", sys.argv[2], "failed", e)
PY
done
done
