#!/bin/bash
# (GPU box) the LDS row window for allele families of several hundred rows: parity (families of 150 / 500), then timing with the
# window on and off (NIMBLE_LDS_WINDOW), families of 100 (register window either way) beside them
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3win}
mkdir -p $OUT
[ -n "$SKIP_PYTEST" ] || timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
[ -n "$SKIP_PYTEST" ] || tail -2 $OUT/pytest.log
for cfg in ${CFGS:-2000_500_2000000 2000_250_2000000 1000_100_4000000}; do cfg=${cfg//_/ }
  for W in ${WS:-1 0}; do
    echo "== T F N = $cfg  NIMBLE_LDS_WINDOW=$W" | tee -a $OUT/fam.txt
    NIMBLE_LDS_WINDOW=$W timeout -k 10 300 python tools/family_one.py $cfg 2>&1 | grep -E "rep [123]|counters" | cut -c1-400 | tee -a $OUT/fam.txt || exit 1
  done
done
