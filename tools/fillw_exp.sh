#!/bin/bash
# Run ON THE GPU BOX: time of denovo_fillw_kernel's first launch under experiment flags.  Usage: tools/fillw_exp.sh <tag> <flags...>
TAG=$1; shift
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
cd /tmp
for F in "$@"; do
  export MCHAP_HIP_FLAGS=$F
  rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_f$F -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_f$F.log 2>&1
  echo "flags $F"; python3 /root/repo/tools/trace_rows.py $OUT/${TAG}_f$F denovo | head -4
  rm -rf $OUT/${TAG}_f$F
done
