#!/bin/bash
# usage: tools/r4_env_ab.sh <tag> VAR "<values>"   (GPU box) bench steps under values of one environment knob, in turns
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--steps 20 --warmup 4 --e2e-reads 0 --cpu-sample 0 --packed-input 0"
for rep in 1 2; do for v in $3; do
  F=$OUT/$2.$v.$rep.log
  env $2=$v timeout -k 10 300 python3 bench.py $ARGS > $F 2>&1
  echo "$2=$v: $(tail -1 $F | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["stage_ms"])')" | tee -a $OUT/summary.txt
done; done
