#!/bin/bash
# k_align against the tile size (development aid; rebuilds the device library on the GPU box for each size)
cd nimble-aligner_amd || exit 1
for B in ${BLOCKS:-128 512 256}; do
  rm -f build/kernels.o
  make HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DNIMBLE_ALIGN_BLOCK=$B $EXTRA_DEFS" > /dev/null 2>&1 || { echo "build failed for $B"; exit 1; }
  echo "== ALIGN_BLOCK $B $EXTRA_DEFS"
  (cd .. && for c in ${CASES:-0 1}; do MIX_CASE=$c timeout -k 10 120 python tools/mix_probe.py 2>&1 | grep k_align; done)
done
