cd /root/repo
mkdir -p gpurun_out/r03c
export GPU_MAX_HW_QUEUES=16
python tools/ploidy_sweep.py 4 > gpurun_out/r03c/k4_lpr.txt 2>&1
python tools/ploidy_sweep.py 6 > gpurun_out/r03c/k6_lpr.txt 2>&1
python tools/config5_once.py 256 2 > gpurun_out/r03c/c5_lpr.txt 2>&1
grep -h -v amdgpu gpurun_out/r03c/k*_lpr.txt gpurun_out/r03c/c5_lpr.txt
timeout 600 python -m pytest tests/test_gpu_fill.py -q -x > gpurun_out/r03c/pytest_fill.txt 2>&1
tail -15 gpurun_out/r03c/pytest_fill.txt
timeout 2000 python -m pytest tests -m gpu -q -x > gpurun_out/r03c/pytest.txt 2>&1
tail -15 gpurun_out/r03c/pytest.txt
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r03c/bench.txt 2>&1
tail -1 gpurun_out/r03c/bench.txt | cut -c1-300
