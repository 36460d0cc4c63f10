#!/bin/bash
# usage: tools/r4_rehearsal_ab.sh <tag>   (GPU box) one-rank rehearsal of the multi-GPU step under a few settings
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--force-sharded --steps 20 --warmup 4 --e2e-reads 0 --cpu-sample 0 --packed-input 0 --form sharded"
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py $ARGS > $OUT/$tag.log 2>&1; echo "$tag: $(tail -1 $OUT/$tag.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("parity_on_union"))')" | tee -a $OUT/summary.txt; }
run base A=1
run aside1280 NIMBLE_DEDUP_ASIDE_SHARDED=1280
run aside640 NIMBLE_DEDUP_ASIDE_SHARDED=640
run aside2048_100 NIMBLE_DEDUP_ASIDE_SHARDED=2048 NIMBLE_ALIGN_GRID_PCT=100
run aside1280_100 NIMBLE_DEDUP_ASIDE_SHARDED=1280 NIMBLE_ALIGN_GRID_PCT=100
run base2 A=1
ARGS="--force-sharded --steps 20 --warmup 4 --e2e-reads 0 --cpu-sample 0 --packed-input 0 --form local"
run local A=1
