#!/bin/bash
# Measurement aid: sampler time of real pileups / moving chains against the size of a chain's likelihood cache in the workspace
# (MCHAP_HIP_CACHE_SLOTS).  Usage (GPU box): bash tools/cache_slots_exp.sh <tag>
tag=${1:-exp}
out=gpurun_out/${tag}_cache_slots.txt
: > $out
for s in "" 4096 16384 65536; do
  echo "== MCHAP_HIP_CACHE_SLOTS=$s" >> $out
  MCHAP_HIP_CACHE_SLOTS=$s python tools/config1_once.py 1,1,16 x >> $out 2>&1
  MCHAP_HIP_CACHE_SLOTS=$s python tools/moving_once.py 16 10000 2>&1 | tail -1 >> $out
  MCHAP_HIP_CACHE_SLOTS=$s python tools/moving_once.py 40 10000 2>&1 | tail -1 >> $out
  MCHAP_HIP_CACHE_SLOTS=$s python tools/e2e_once.py 1000 2 32768 >> $out 2>&1
done
cat $out
