cd /root/repo
mkdir -p gpurun_out/r03g
export GPU_MAX_HW_QUEUES=16
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r03g/pytest.txt 2>&1
tail -5 gpurun_out/r03g/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/r03g/bench.json 2> gpurun_out/r03g/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03g/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
e=j["extra"]
print("config1", e["config1"]["value"], e["config1"]["sampler_ms"]); print("dedup", e["config2_dedup"]["value"]); print("moving", e["moving"]["reads_16"]["value"], e["moving"]["reads_40"]["value"]); print("c4", e["config4"]["value"], "c5", e["config5"]["value"])
PY
