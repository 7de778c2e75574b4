#!/bin/bash
# Run ON THE GPU BOX: per-launch times of one sampler call with the table completion as its own kernel (default) and inside the
# exporting launch (MCHAP_HIP_FLAGS=1024).  Usage: tools/fillw_trace.sh <tag>
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_new -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_new.log 2>&1
export MCHAP_HIP_FLAGS=1024
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_old -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_old.log 2>&1
unset MCHAP_HIP_FLAGS
python3 /root/repo/tools/trace_rows.py $OUT/${TAG}_new denovo > $OUT/${TAG}_new_rows.txt
python3 /root/repo/tools/trace_rows.py $OUT/${TAG}_old denovo > $OUT/${TAG}_old_rows.txt
rm -rf $OUT/${TAG}_new $OUT/${TAG}_old
echo new; cat $OUT/${TAG}_new_rows.txt; echo old; cat $OUT/${TAG}_old_rows.txt
