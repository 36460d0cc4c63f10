#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> <python script + args...>   (runs on the GPU box)
# One rocprofv3 --pmc pass per counter group (never combined with sys/hip trace), summarised per kernel.
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  echo "pass $i: $group" >> $OUT/progress.txt
  timeout -k 5 150 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- python3 "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed" >> $OUT/progress.txt
done <<GROUPS
${PMC_GROUPS}
GROUPS
python3 - $OUT <<'PY'
import csv, glob, sys, re, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        if not m or "nimble" not in r["Kernel_Name"]: continue
        agg[m.group(1) + (m.group(2) or "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as g:
    for k in sorted(agg):
        g.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            g.write("  %-36s avg/dispatch %.6g  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
