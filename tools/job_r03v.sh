cd /root/repo
mkdir -p gpurun_out/r03v
export GPU_MAX_HW_QUEUES=16
export SWEEP_KERNELS=3,0
for R in 8 16 30 40 60; do python tools/hard_sweep.py 10000 $R 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r03v/sweep.txt
cat gpurun_out/r03v/sweep.txt
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03v/bench.json 2> gpurun_out/r03v/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03v/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
PY
python tools/example_by_locus.py 2>&1 | grep -v amdgpu.ids | tail -8
