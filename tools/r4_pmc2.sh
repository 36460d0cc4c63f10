#!/bin/bash
# usage: tools/r4_pmc2.sh <outdir> <cases...>   (GPU box) memory-path counters of the k_align launches (TA / TCP / UTCL1 / TCC)
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
PMCG=("TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr" \
        "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
        "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum" \
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
        "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
        "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU")
for c in "$@"; do
  i=0
  for g in "${PMCG[@]}"; do
    MIX_CASE=$c timeout -k 10 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/c${c}_g$i -- python3 tools/mix_probe.py > $OUT/c${c}_g$i.log 2>&1
    i=$((i+1))
  done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.OrderedDict()
for d in sorted(glob.glob(out + "/c*_g*/")):
    case = d.split("/")[-2].split("_")[0]
    rows = {}
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_align" not in r["Kernel_Name"]: continue
            m = re.search(r"k_align<[^>]*>", r["Kernel_Name"])
            rows.setdefault((m.group(0) if m else "k_align", int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    last = {}
    for k in sorted(rows): last[k[0]] = k
    for name, k in last.items():
        acc.setdefault((case, name), {}).update(rows[k])
for (case, name), v in acc.items():
    print(case, name)
    for kv in sorted(v.items()): print("    %-40s %.4g" % kv)
PY
find $OUT -name "*.csv" -size +100k -delete
