#!/bin/bash
# Run ON THE GPU BOX: the configs[4] part of profile_round.sh alone (three SQ counter groups, each in its own run, and the kernel
# trace with --stats of tools/config5_once.py 256 1).  Usage: tools/profile_config5.sh <tag> -> gpurun_out/<tag>_c5_*
TAG=${1:-r04b}
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64"
SQ3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32"
for G in 1 2 3; do
  eval C=\$SQ$G
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_c5_sq$G -- python3 /root/repo/tools/config5_once.py 256 1 > $OUT/${TAG}_c5_sq$G.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c5_trace -- python3 /root/repo/tools/config5_once.py 256 1 > $OUT/${TAG}_c5_trace.log 2>&1
python3 /root/repo/tools/pmc_sq.py $OUT/${TAG}_c5_sq1 $OUT/${TAG}_c5_sq2 $OUT/${TAG}_c5_sq3 > $OUT/${TAG}_c5_sq.json 2>> $OUT/${TAG}_c5_sq1.log
find $OUT/${TAG}_c5_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_c5_kernel_stats.csv \;
rm -rf $OUT/${TAG}_c5_sq1 $OUT/${TAG}_c5_sq2 $OUT/${TAG}_c5_sq3 $OUT/${TAG}_c5_trace
ls $OUT | grep ${TAG}_c5 || true
