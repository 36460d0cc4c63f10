#!/bin/bash
# clock check (development aid): GPU cycles vs wall time of k_align with 1 and 2 calls in flight
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for d in 1 2; do
timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/d$d -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-reads 0 --read-sets 1 --depth $d > $OUT/d$d.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("d1", "d2"):
    cyc = collections.defaultdict(list); dur = collections.defaultdict(list)
    for f in glob.glob(out + "/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "nimble" not in r["Kernel_Name"] or r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
            k = r["Kernel_Name"].split("(")[0][-28:]
            cyc[k].append(float(r["Counter_Value"]))
    for f in glob.glob(out + "/%s/*/*kernel_trace.csv" % d):
        for r in csv.DictReader(open(f)):
            if "nimble" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("(")[0][-28:]
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k in cyc:
        c = sum(cyc[k][3:]) / max(len(cyc[k][3:]), 1); t = sum(dur[k][3:]) / max(len(dur[k][3:]), 1)
        print(d, k, "cycles %.4g  ns %.4g  GHz %.3f" % (c, t, c / t if t else 0))
PY
