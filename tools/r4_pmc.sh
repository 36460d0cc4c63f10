#!/bin/bash
# usage: tools/r4_pmc.sh <outdir> <cases...>   (GPU box) SQ counters of the k_align launches per read class, fast launch on and off
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
for f in 1 0; do
for c in "$@"; do
  g="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY"
  NIMBLE_FAST_ALIGN=$f MIX_CASE=$c timeout -k 10 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/f${f}_c${c} -- python3 tools/mix_probe.py > $OUT/f${f}_c${c}.log 2>&1
done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
for d in sorted(glob.glob(out + "/f*_c*/")):
    rows = collections.OrderedDict()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_align" not in r["Kernel_Name"]: continue
            m = re.search(r"k_align<[^>]*>", r["Kernel_Name"])
            rows.setdefault((int(r["Dispatch_Id"]), m.group(0) if m else "k_align"), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    keys = sorted(rows)
    last = {}
    for k in keys: last[k[1]] = k      # the last dispatch of each kernel (counters off)
    for name, k in sorted(last.items()):
        v = rows[k]
        util = v.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * v.get("SQ_ACTIVE_INST_VALU", 1))
        print(d.split("/")[-2], name, " ".join("%s=%.4g" % kv for kv in sorted(v.items())), "lane_use=%.3f" % util)
PY
find $OUT -name "*.csv" -size +100k -delete
