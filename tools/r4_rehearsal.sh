#!/bin/bash
# usage: tools/r4_rehearsal.sh <tag>   (GPU box) the multi-GPU step rehearsed on one rank over real RCCL: bench line + kernel trace
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--force-sharded --steps 20 --warmup 4 --e2e-reads 0 --cpu-sample 0 --packed-input 0"
timeout -k 10 300 python3 bench.py $ARGS ${FORM:+--form $FORM} > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-400
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS --form sharded > $OUT/trace.log 2>&1
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY' | tee $OUT/kernels.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-70s calls %6s  avg %9.1f us  total %9.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
