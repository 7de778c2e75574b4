#!/bin/bash
# Run ON THE GPU BOX: the launches of docs/example through the program (tools/config1_once.py, 440 units) in time order.
TAG=${1:-r04_c1}
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG} -- python3 /root/repo/tools/config1_once.py ${2:-1} only > $OUT/${TAG}.log 2>&1
python3 /root/repo/tools/trace_rows.py $OUT/${TAG} mchap > $OUT/${TAG}_rows.txt
rm -rf $OUT/${TAG}
cat $OUT/${TAG}_rows.txt | cut -c1-160
