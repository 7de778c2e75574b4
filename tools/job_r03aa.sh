cd /root/repo
mkdir -p gpurun_out/r03aa
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r03aa/pytest.txt 2>&1
tail -5 gpurun_out/r03aa/pytest.txt
python bench.py --no-cpu-baseline > gpurun_out/r03aa/bench.json 2> gpurun_out/r03aa/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03aa/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
e=j["extra"]
print("config1", e["config1"]["value"], e["config1"]["sampler_ms"], "at_scale", e["config1"]["at_scale"]["value"])
print("dedup", e["config2_dedup"]["value"]); print("moving", e["moving"]["reads_16"]["value"], e["moving"]["reads_40"]["value"]); print("c4", e["config4"]["value"], "c5", e["config5"]["value"], e["config5"]["value_one_in_flight"], "e2e", e["program_e2e"]["value"])
PY
