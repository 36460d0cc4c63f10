#!/bin/bash
# usage: tools/r4_rehearsal_libs.sh <tag> "<variants>"   (GPU box) one-rank rehearsal of the multi-GPU step across library builds
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--force-sharded --steps 20 --warmup 4 --e2e-reads 0 --cpu-sample 0 --packed-input 0 --form sharded"
for v in $2; do
  D=$PWD/nimble-aligner_amd/libv/$v; [ "$v" = lib ] && D=$PWD/nimble-aligner_amd/lib
  F=$OUT/$v.$RANDOM.log
  NIMBLE_LIB_DIR=$D timeout -k 10 300 python3 bench.py $ARGS > $F 2>&1
  echo "$v: $(tail -1 $F | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("parity_on_union"))')" | tee -a $OUT/summary.txt
done
