#!/bin/bash
# Run ON THE GPU BOX: per-launch times of one pass of BASELINE.json configs[4] (256 loci).  Usage: tools/config5_trace.sh <tag>
TAG=${1:-r04_c5}
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG} -- python3 /root/repo/tools/config5_once.py 256 1 > $OUT/${TAG}.log 2>&1
python3 /root/repo/tools/trace_rows.py $OUT/${TAG} denovo > $OUT/${TAG}_rows.txt
rm -rf $OUT/${TAG}
cat $OUT/${TAG}_rows.txt
