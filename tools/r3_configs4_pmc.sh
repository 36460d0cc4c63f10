#!/bin/bash
# usage: tools/r3_configs4_pmc.sh <tag>   (GPU box) L2 / fabric counters of k_align on BASELINE.json configs[4] (80 M reads, 5 k
# features) beside configs[2], one --pmc pass per counter group
export TMPDIR=/tmp
TAG=${1:-r3_c4}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for wl in configs4 configs2; do
  PMC_GROUPS=$'TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum\nFETCH_SIZE\nWRITE_SIZE\nSQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU' \
    bash tools/pmc_passes.sh $OUT/$wl bench.py --workload $wl --steps 3 --warmup 1 --cpu-sample 0 --e2e-reads 0 --read-sets 1 --packed-input 0 > $OUT/$wl.txt 2>&1
  grep -A 16 "k_align<false, false, false>" $OUT/$wl/summary.txt | head -18
done
