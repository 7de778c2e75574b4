#!/bin/bash
# Measurement aid: phase timers of the sampler on the docs/example targets whose chains keep moving.  Usage: bash tools/stats_example.sh <tag>
tag=${1:-exp}
out=gpurun_out/${tag}_stats_example.txt
: > $out
export MCHAP_HIP_LIB=mchap_amd/csrc/libmchap_hip_phases.so
for l in locus015 locus012 locus005 locus001 locus018; do
  python tools/stats_example.py $l >> $out 2>&1
done
MCHAP_HIP_CACHE_SLOTS=65536 python tools/stats_example.py locus015 >> $out 2>&1
cat $out
