cd /root/repo
mkdir -p gpurun_out/r03ae
export GPU_MAX_HW_QUEUES=16
export SWEEP_KERNELS=3,0
for R in 8 16 30 40 60; do python tools/hard_sweep.py 10000 $R 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03ae/sweep.txt
grep phased gpurun_out/r03ae/sweep.txt; grep -c "same traces" gpurun_out/r03ae/sweep.txt
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03ae/bench.json 2> gpurun_out/r03ae/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03ae/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
PY
python tools/example_by_locus.py 2>&1 | grep -v amdgpu.ids | tail -3
python tools/config5_once.py 256 2 2>&1 | grep -v amdgpu | tail -1
timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_denovo.py tests/test_gpu_example.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q -x > gpurun_out/r03ae/pytest.txt 2>&1; tail -3 gpurun_out/r03ae/pytest.txt
