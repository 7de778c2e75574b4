#!/bin/bash
# Run ON THE GPU BOX: the launches of one `assemble` job (tools/e2e_once.py) in time order.  Usage: tools/e2e_trace.sh <tag> [loci]
TAG=${1:-r04_e2e}
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG} -- python3 /root/repo/tools/e2e_once.py ${2:-1000} 1 > $OUT/${TAG}.log 2>&1
python3 /root/repo/tools/trace_rows.py $OUT/${TAG} "" > $OUT/${TAG}_rows.txt
rm -rf $OUT/${TAG}
head -40 $OUT/${TAG}_rows.txt
