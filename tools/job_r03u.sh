cd /root/repo
mkdir -p gpurun_out/r03u
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests/test_gpu_posterior.py tests/test_gpu_example.py tests/test_gpu_fuzz.py tests/test_gpu_assemble_goldens.py tests/test_gpu_cli.py -q -x > gpurun_out/r03u/pytest.txt 2>&1; tail -8 gpurun_out/r03u/pytest.txt
python tools/profile_assemble_host.py 16 2>&1 | grep -v amdgpu.ids | cut -c1-170 > gpurun_out/r03u/profile_host.txt; head -30 gpurun_out/r03u/profile_host.txt
