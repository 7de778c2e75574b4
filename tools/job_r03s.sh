cd /root/repo
mkdir -p gpurun_out/r03s
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_denovo.py tests/test_gpu_example.py tests/test_gpu_fuzz.py tests/test_gpu_assemble_goldens.py -q -x > gpurun_out/r03s/pytest.txt 2>&1; tail -8 gpurun_out/r03s/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/r03s/bench.json 2> gpurun_out/r03s/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03s/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
e=j["extra"]
print("config1", json.dumps(e["config1"])[:1500])
print("moving", e["moving"]["reads_16"]["value"], e["moving"]["reads_40"]["value"], "dedup", e["config2_dedup"]["value"], "c5", e["config5"]["value"], "e2e", e["program_e2e"]["value"])
PY
