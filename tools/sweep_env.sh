#!/bin/bash
# usage: tools/sweep_env.sh <outdir> <VAR> <v1> <v2> ...   (GPU box) -- tools/align_ab.py under each value of one env var
OUT=$1; VAR=$2; shift 2; mkdir -p $OUT
for v in "$@"; do
  env $VAR=$v timeout -k 10 150 python tools/align_ab.py ${AB_T:-1000} ${AB_N:-10000000} 4 > $OUT/${VAR}_$v.log 2>&1 || exit 1
  echo "$VAR=$v $(grep '^rep 3' $OUT/${VAR}_$v.log | cut -c1-160)" | tee -a $OUT/summary.txt
done
