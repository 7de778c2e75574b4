cd /root/repo
mkdir -p gpurun_out/r03a
export GPU_MAX_HW_QUEUES=16
for lib in stats; do
  export MCHAP_HIP_LIB=/root/repo/mchap_amd/csrc/libmchap_hip_$lib.so
  python tools/stats_run.py 10000 1000 0 200 > gpurun_out/r03a/${lib}_cfg2.txt 2>&1
  MCHAP_HIP_PIPE_STOP=1 python tools/stats_run.py 10000 1000 0 200 > gpurun_out/r03a/${lib}_cfg2_stop.txt 2>&1
  STATS_K=8 STATS_M=20 STATS_CHAINS=4 python tools/stats_run.py 256 2000 0 1000 > gpurun_out/r03a/${lib}_cfg5.txt 2>&1
  python tools/stats_run.py 10000 1000 0 16 > gpurun_out/r03a/${lib}_moving16.txt 2>&1
  STATS_K=6 python tools/stats_run.py 10000 1000 0 200 > gpurun_out/r03a/${lib}_k6.txt 2>&1
done
