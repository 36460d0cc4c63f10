#!/bin/bash
# per-read-class L2/fabric counters of k_align (development aid): tools/pmc_ablate.sh <outdir>
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
timeout -k 5 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/p1 -- python3 tools/ablate_reads.py > $OUT/p1.log 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/p2 -- python3 tools/ablate_reads.py > $OUT/p2.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    rows = collections.OrderedDict()
    for f in glob.glob(out + "/%s/*/*counter_collection.csv" % p):
        for r in csv.DictReader(open(f)):
            if "k_align" not in r["Kernel_Name"]: continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d in sorted(rows):
        print(p, d, " ".join("%s=%.4g" % kv for kv in sorted(rows[d].items())))
PY
