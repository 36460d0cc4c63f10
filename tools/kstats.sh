#!/bin/bash
# usage: tools/kstats.sh <outdir> <bench args...>   kernel-trace stats (+ a trace csv) of one bench command, on the GPU box
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py "$@" > $OUT/run.log 2>&1 || exit 2
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in list(csv.reader(open(f)))[:22]:
        print(",".join(r[:5])[:200])
PY
