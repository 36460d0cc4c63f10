#!/bin/bash
# usage: tools/r3_instr_budget.sh <tag>  (GPU box) VALU / SALU / VMEM instruction counts and L2 counters of k_align per read class
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3_budget}; mkdir -p $OUT
for c in 0 1 2 4 7 9 10; do
  for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    n=$(echo $g | cut -c1-6)
    MIX_CASE=$c timeout -k 10 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/c${c}_$n -- python3 tools/mix_probe.py > $OUT/c${c}_$n.log 2>&1
  done
  grep k_align $OUT/c${c}_SQ_INS.log | head -1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/c*_*/")):
    rows = collections.OrderedDict()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_align" not in r["Kernel_Name"]: continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(rows)[-1:]:
        print(d.split("/")[-2], " ".join("%s=%.4g" % kv for kv in sorted(rows[k].items())))
PY
find $OUT -name "*.csv" -size +200k -delete
