#!/bin/bash
# usage: tools/r4_tail_ab.sh <tag>   (GPU box) bench steps with the call's tail beside the next call's pack (as is) or align
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--steps 20 --warmup 4 --e2e-reads 0 --cpu-sample 0 --packed-input 0"
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py $ARGS > $OUT/$tag.log 2>&1; echo "$tag: $(tail -1 $OUT/$tag.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["stage_ms"])')" | tee -a $OUT/summary.txt; }
run base A=1
run beside100 NIMBLE_TAIL_BESIDE_ALIGN=1
run beside86 NIMBLE_TAIL_BESIDE_ALIGN=1 NIMBLE_ALIGN_GRID=86
run beside72 NIMBLE_TAIL_BESIDE_ALIGN=1 NIMBLE_ALIGN_GRID=72
run beside86_g256 NIMBLE_TAIL_BESIDE_ALIGN=1 NIMBLE_ALIGN_GRID=86 NIMBLE_DEDUP_ASIDE=256
run beside86_g4096 NIMBLE_TAIL_BESIDE_ALIGN=1 NIMBLE_ALIGN_GRID=86 NIMBLE_DEDUP_ASIDE=4096
run base2 A=1
