#!/bin/bash
# usage: tools/r4_configs3_libs.sh <tag> "<variants>"   (GPU box) configs[3] (paired) across library builds
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
for v in $2; do
  D=$PWD/nimble-aligner_amd/libv/$v; [ "$v" = lib ] && D=$PWD/nimble-aligner_amd/lib
  F=$OUT/$v.$RANDOM.log
  NIMBLE_LIB_DIR=$D timeout -k 10 300 python3 bench.py --workload ${WL:-families100} --cpu-sample 0 --e2e-reads 0 > $F 2>&1
  echo "$v: $(tail -1 $F | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"])')" | tee -a $OUT/summary.txt
done
