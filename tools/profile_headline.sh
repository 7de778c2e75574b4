#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: the configs[1] part of tools/profile_round.sh alone -- kernel trace of the
# default bench command (4 and 1 passes in flight), then the PMC passes of one sampler call (each counter group in its own run).
# Usage: tools/profile_headline.sh <tag> -> gpurun_out/<tag>_*
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out
mkdir -p $OUT
[ -f /root/repo/tools/libpmc_calib.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /root/repo/tools/libpmc_calib.so /root/repo/tools/pmc_calib.hip
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 /root/repo/bench.py --no-cpu-baseline --no-extras > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace1 -- python3 /root/repo/bench.py --inflight 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_trace1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_cal_fetch -- python3 /root/repo/tools/pmc_calib_run.py > $OUT/${TAG}_cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_cal_write -- python3 /root/repo/tools/pmc_calib_run.py > $OUT/${TAG}_cal_write.log 2>&1
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64"
SQ3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32"
rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq1 -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_pmc_sq1.log 2>&1
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq2 -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_pmc_sq2.log 2>&1
rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq3 -- python3 /root/repo/tools/prof_run.py full 10000 > $OUT/${TAG}_pmc_sq3.log 2>&1
python3 /root/repo/tools/pmc_sq.py $OUT/${TAG}_pmc_sq1 $OUT/${TAG}_pmc_sq2 $OUT/${TAG}_pmc_sq3 > $OUT/${TAG}_c2_sq.json 2>> $OUT/${TAG}_pmc_sq1.log
ls $OUT | grep $TAG
