#!/bin/bash
# usage: tools/kstats.sh <outdir> <bench args...>   kernel-trace stats (+ a trace csv) of one bench command, on the GPU box
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py "$@" > $OUT/run.log 2>&1 || exit 2
python3 - $OUT <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:18]:
        name = re.sub(r"\(.*", "", r["Name"].replace("nimble::(anonymous namespace)::", "").replace("void ", ""))
        print("%-48s calls %5s  avg %9.1f us  %5.1f %%" % (name[:48], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       float(r["Percentage"])))
PY
grep -o '"ms_per_step": [0-9.]*' $OUT/run.log
