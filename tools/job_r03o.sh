cd /root/repo
mkdir -p gpurun_out/r03o
export GPU_MAX_HW_QUEUES=16
export SWEEP_KERNELS=3,0
for R in 8 16 30 40; do python tools/hard_sweep.py 10000 $R 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03o/sweep.txt
cat gpurun_out/r03o/sweep.txt
for L in "" _u2 _u4; do
  MCHAP_HIP_LIB=/root/repo/mchap_amd/csrc/libmchap_hip$L.so python tools/exact_bench.py > gpurun_out/r03o/exact$L.txt 2>&1; tail -2 gpurun_out/r03o/exact$L.txt
done
timeout 900 python -m pytest tests/test_gpu_denovo.py tests/test_gpu_fuzz.py tests/test_gpu_exact.py -q -x > gpurun_out/r03o/pytest.txt 2>&1; tail -3 gpurun_out/r03o/pytest.txt
timeout 900 python tests/fuzz_ragged.py 40 2 1500 44 > gpurun_out/r03o/fuzz_ragged2.txt 2>&1; tail -n 4 gpurun_out/r03o/fuzz_ragged2.txt
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03o/bench.json 2> gpurun_out/r03o/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03o/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
PY
