#!/bin/bash
# usage: tools/r3_family_pmc.sh <tag>   (GPU box) what bounds k_align on allele families: counters of the WIDE instantiation on
# families of 100 (register window) and 500 (LDS window), one --pmc pass per group
export TMPDIR=/tmp
TAG=${1:-r3_fpmc}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in "1000 100 4000000" "2000 500 2000000"; do
  d=$OUT/f$(echo $cfg | cut -d' ' -f2)
  PMC_GROUPS=$'TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum\nSQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU\nSQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE\nTCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum' \
    bash tools/pmc_passes.sh $d tools/family_one.py $cfg > $d.txt 2>&1
  grep -A 30 "k_align" $d/summary.txt | head -40
done
