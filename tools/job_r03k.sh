cd /root/repo
mkdir -p gpurun_out/r03k
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests/test_gpu_example.py tests/test_gpu_cli.py tests/test_gpu_assemble_goldens.py tests/test_gpu_call_exact_goldens.py -q -x > gpurun_out/r03k/pytest.txt 2>&1
tail -5 gpurun_out/r03k/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/r03k/bench.json 2> gpurun_out/r03k/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03k/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
e=j["extra"]
print("config1", json.dumps(e["config1"])[:900]); print("e2e", json.dumps(e["program_e2e"])[:900])
print("c4", e["config4"]["roofline"]); print("c5", e["config5"]["roofline"])
PY
