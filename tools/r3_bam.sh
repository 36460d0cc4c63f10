#!/bin/bash
# (GPU box) the BAM pipeline end to end: parity tests, then lib/nimble on a synthetic 10x-style BAM at several gzip levels
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3bam}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_bam.py tests/test_gpu_umi.py -q -m gpu -x > $OUT/pytest.log 2>&1 || { grep -v Unpaired $OUT/pytest.log | tail -30; exit 1; }
grep -v Unpaired $OUT/pytest.log | tail -1
for lv in ${LEVELS:-6 4 1}; do
  echo "== NIMBLE_GZIP_LEVEL=$lv" | tee -a $OUT/out.txt
  NIMBLE_GZIP_LEVEL=$lv NIMBLE_HOST_TIMING=1 timeout -k 10 800 python tools/e2e_bam.py ${PAIRS:-4000000} 8 2>&1 | grep -E "run|nimble host" | cut -c1-500 | tee -a $OUT/out.txt
done
