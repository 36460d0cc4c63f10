#!/bin/bash
# usage: tools/r3_step.sh <tag>   (GPU box) parity subset, then k_align A/B: local re-seed on/off, by read class, with L2 counters
export TMPDIR=/tmp
TAG=${1:-r3a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_adversarial.py tests/test_gpu_pipeline.py -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
for LR in 1 0; do
  for c in 0 2 1 4; do
    NIMBLE_LOCAL_RESEED=$LR MIX_CASE=$c timeout -k 10 200 python tools/mix_probe.py 2>&1 | grep k_align | sed "s/^/LR=$LR /" | tee -a $OUT/mix.txt
  done
done
for LR in 1 0; do
  for c in 0 2; do
    NIMBLE_LOCAL_RESEED=$LR MIX_CASE=$c timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT/pmc_lr${LR}_c$c -- python3 tools/mix_probe.py > $OUT/pmc_lr${LR}_c$c.log 2>&1
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
    for k in sorted(rows)[-2:]:
        print(d.split("/")[-2], k, " ".join("%s=%.4g" % kv for kv in sorted(rows[k].items())))
PY
