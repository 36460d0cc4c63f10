#!/bin/bash
# usage: tools/r4_pmc_cases.sh <tag> "<cases>"   (GPU box) VALU instructions and lane use of k_align per kind of read (tools/mix_probe.py cases)
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
for c in ${2:-0 1 2 4 9}; do
  MIX_CASE=$c timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 tools/mix_probe.py > $OUT/pmc_$c.log 2>&1 || exit 2
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc_*/")):
    rows = collections.OrderedDict()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_align" not in r["Kernel_Name"]: continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(rows)[-1:]:
        v = rows[k]
        print(d.split("/")[-2], " ".join("%s=%.4g" % kv for kv in sorted(v.items())), "lane_use=%.3f" % (v["SQ_THREAD_CYCLES_VALU"] / 64 / v["SQ_INSTS_VALU"]))
PY
