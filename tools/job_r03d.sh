cd /root/repo
mkdir -p gpurun_out/r03d
export GPU_MAX_HW_QUEUES=16
export MCHAP_HIP_LIB=/root/repo/mchap_amd/csrc/libmchap_hip_phases.so
python tools/stats_run.py 10000 1000 0 200 > gpurun_out/r03d/phases_cfg2.txt 2>&1
STATS_K=8 STATS_M=20 STATS_CHAINS=4 python tools/stats_run.py 256 2000 0 1000 > gpurun_out/r03d/phases_cfg5.txt 2>&1
STATS_K=6 python tools/stats_run.py 10000 1000 0 200 > gpurun_out/r03d/phases_k6.txt 2>&1
unset MCHAP_HIP_LIB
grep -h "fill kernel\|TOTAL\|^units" gpurun_out/r03d/phases_*.txt
timeout 900 python -m pytest tests/test_gpu_fill.py tests/test_gpu_example.py tests/test_gpu_cli.py -q > gpurun_out/r03d/pytest_new.txt 2>&1
tail -15 gpurun_out/r03d/pytest_new.txt
