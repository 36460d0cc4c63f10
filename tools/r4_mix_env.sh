#!/bin/bash
# usage: tools/r4_mix_env.sh <tag> VAR "<values>" "<cases>"   (GPU box) k_align alone per kind of read under values of one knob
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do for v in $3; do for c in ${4:-0 4}; do
  env $2=$v MIX_CASE=$c timeout -k 10 200 python tools/mix_probe.py 2>&1 | grep k_align | sed "s/^/$2=$v: /" | tee -a $OUT/mix.txt
done; done; done
