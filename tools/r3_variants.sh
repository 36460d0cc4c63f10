#!/bin/bash
# usage: tools/r3_variants.sh <tag> "<variants>" "<cases>"   (GPU box) mix_probe across library builds (libv/<name>; "lib" = the tree's)
export TMPDIR=/tmp
TAG=$1; VARS=${2:-"lib base"}; CASES=${3:-"0 1 2"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for v in $VARS; do
  D=$PWD/nimble-aligner_amd/libv/$v; [ "$v" = lib ] && D=$PWD/nimble-aligner_amd/lib
  for c in $CASES; do
    NIMBLE_LIB_DIR=$D MIX_CASE=$c timeout -k 10 200 python tools/mix_probe.py 2>&1 | grep k_align | sed "s/^/$v: /" | tee -a $OUT/mix.txt
  done
done
if [ -n "$PMC_VARS" ]; then
  for v in $PMC_VARS; do
    D=$PWD/nimble-aligner_amd/libv/$v; [ "$v" = lib ] && D=$PWD/nimble-aligner_amd/lib
    export NIMBLE_LIB_DIR=$D
    for g in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
      n=$(echo $g | cut -c1-6)
      MIX_CASE=${PMC_CASE:-0} timeout -k 10 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/pmc_${v}_$n -- python3 tools/mix_probe.py > $OUT/pmc_${v}_$n.log 2>&1
    done
  done
  python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc_*/")):
    rows = collections.OrderedDict()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_align" not in r["Kernel_Name"]: continue
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(rows)[-1:]:
        print(d.split("/")[-2], " ".join("%s=%.4g" % kv for kv in sorted(rows[k].items())))
PY
fi
