#!/bin/bash
# usage: tools/profile_round.sh <tag>      (runs on the GPU box; writes gpurun_out/<tag>/)
# 1. plain bench (the line the driver records)  2. rocprofv3 --kernel-trace --stats of the same command
# 3. separate --pmc passes (never combined with any trace but --kernel-trace) for HBM traffic and SQ occupancy.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench_n1.json.log 2> $OUT/bench_n1.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 --e2e-reads 0 > $OUT/bench_prof.log 2>&1 || exit 2
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "nimble" in r[0]]
    with open(out + "/kernel_stats_nimble.csv", "w", newline="") as g:
        csv.writer(g).writerows(keep)
    for r in keep:
        print(",".join(r)[:220])
PY
PMC_GROUPS=$'FETCH_SIZE\nWRITE_SIZE\nSQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD\nTCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum' \
  bash tools/pmc_passes.sh $OUT/pmc bench.py --steps 3 --warmup 1 --cpu-sample 0 --e2e-reads 0 --read-sets 1
