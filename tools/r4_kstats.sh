#!/bin/bash
# usage: tools/r4_kstats.sh <outdir> <command...>   kernel-trace stats of any python command on the GPU box (round 4)
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 2; }
python3 - $OUT <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:14]:
        name = r["Name"].replace("nimble::(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(nimble::DevIndex.*", "", name)
        print("%-60s calls %5s  avg %9.1f us  %5.1f %%" % (name[:60], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                       float(r["Percentage"])))
PY
tail -2 $OUT/run.log
