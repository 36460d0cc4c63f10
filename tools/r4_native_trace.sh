#!/bin/bash
# usage: tools/r4_native_trace.sh <tag>   (GPU box) kernel trace of the native multi-GPU step rehearsed on one rank
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--force-sharded --form native --steps 12 --warmup 3 --e2e-reads 0 --cpu-sample 0 --packed-input 0"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv")[0]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:58], r["Queue_Id"]) for r in csv.DictReader(open(f)))
al = [k for k in ks if "k_align" in k[2]]
a, b = al[-5], al[-4]
print("step %.3f ms" % ((b[0] - a[0]) / 1e6))
prev_end = {}
for k in ks:
    if a[0] - 30000 <= k[0] < b[0] - 30000:
        gap = (k[0] - prev_end.get(k[3], k[0])) / 1e6
        print("%9.3f %8.3f gap %7.3f  %-58s q=%s" % ((k[0] - a[0]) / 1e6, (k[1] - k[0]) / 1e6, gap, k[2], k[3]))
    prev_end[k[3]] = max(prev_end.get(k[3], 0), k[1])
PY
