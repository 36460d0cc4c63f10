#!/bin/bash
# usage: tools/r4_final_lines.sh <tag>   (GPU box) the bench lines beside the default one: configs[3], configs[4], allele families,
# the multi-GPU step rehearsed on one rank (torch and native form), and configs[4]'s L2 counters
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
C="--cpu-sample 0 --e2e-reads 0"
timeout -k 10 300 python3 bench.py --workload configs3 --steps 6 --warmup 2 $C > $OUT/configs3.json.log 2> $OUT/configs3.err || exit 2
timeout -k 10 400 python3 bench.py --workload configs4 --steps 6 --warmup 2 $C > $OUT/configs4.json.log 2> $OUT/configs4.err || exit 3
timeout -k 10 300 python3 bench.py --workload families100 $C > $OUT/families100.json.log 2> $OUT/families100.err || exit 4
timeout -k 10 300 python3 bench.py --workload families500 $C > $OUT/families500.json.log 2> $OUT/families500.err || exit 5
timeout -k 10 300 python3 bench.py --force-sharded --form sharded --steps 10 --warmup 3 $C > $OUT/torch_w1.json.log 2> $OUT/torch_w1.err || exit 6
timeout -k 10 300 python3 bench.py --force-sharded --form native --steps 10 --warmup 3 $C > $OUT/native_w1.json.log 2> $OUT/native_w1.err || exit 7
for f in configs3 configs4 families100 families500 torch_w1 native_w1; do
  echo "$f: $(tail -1 $OUT/$f.json.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}; print(d["value"], d["ms_per_step"], r.get("kernel_ms"), r.get("frac"), d.get("parity_on_union"))')" | tee -a $OUT/summary.txt
done
PMC_GROUPS=$'TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum\nFETCH_SIZE\nWRITE_SIZE' \
  bash tools/pmc_passes.sh $OUT/configs4_pmc bench.py --workload configs4 --steps 3 --warmup 1 --cpu-sample 0 --e2e-reads 0 --read-sets 1 --packed-input 0 > $OUT/configs4_pmc.txt 2>&1
grep -A 8 "k_align<false, false, false" $OUT/configs4_pmc/summary.txt | head -12 | tee -a $OUT/summary.txt
