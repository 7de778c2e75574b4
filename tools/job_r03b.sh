cd /root/repo
mkdir -p gpurun_out/r03b
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b/pytest.txt 2>&1
tail -5 gpurun_out/r03b/pytest.txt
export MCHAP_HIP_LIB=/root/repo/mchap_amd/csrc/libmchap_hip_phases.so
python tools/stats_run.py 10000 1000 0 200 > gpurun_out/r03b/phases_cfg2.txt 2>&1
STATS_K=8 STATS_M=20 STATS_CHAINS=4 python tools/stats_run.py 256 2000 0 1000 > gpurun_out/r03b/phases_cfg5.txt 2>&1
python tools/stats_run.py 10000 1000 0 16 > gpurun_out/r03b/phases_moving16.txt 2>&1
STATS_K=6 python tools/stats_run.py 10000 1000 0 200 > gpurun_out/r03b/phases_k6.txt 2>&1
unset MCHAP_HIP_LIB
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r03b/bench.txt 2>&1
MCHAP_HIP_FLAGS=32 MCHAP_HIP_PIPE_STOP=1 python tools/ploidy_sweep.py 4 > gpurun_out/r03b/nofill_k4.txt 2>&1
MCHAP_HIP_PIPE_STOP=1 python tools/ploidy_sweep.py 4 > gpurun_out/r03b/stop_k4.txt 2>&1
python tools/ploidy_sweep.py 4 > gpurun_out/r03b/full_k4.txt 2>&1
grep -v amdgpu gpurun_out/r03b/*_k4.txt
