#!/bin/bash
# Measurement aid: phase timers (and, with the `stats` build, event counters) of the sampler on the docs/example targets whose
# chains keep moving.  Usage: bash tools/stats_example.sh <tag> [library]
tag=${1:-exp}
out=gpurun_out/${tag}_stats_example.txt
: > $out
export MCHAP_HIP_LIB=${2:-mchap_amd/csrc/libmchap_hip_phases.so}
for l in locus015 locus012 locus005 locus001 locus018; do
  python tools/stats_example.py $l >> $out 2>&1
done
MCHAP_HIP_CACHE_SLOTS=1024 python tools/stats_example.py locus015 >> $out 2>&1
cat $out
