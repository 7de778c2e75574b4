cd /root/repo
mkdir -p gpurun_out/r03y
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r03y/pytest.txt 2>&1
tail -5 gpurun_out/r03y/pytest.txt
python bench.py > gpurun_out/r03y/bench.json 2> gpurun_out/r03y/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03y/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"], "h2d", j.get("value_incl_h2d"))
e=j["extra"]
print("config1", e["config1"]["value"], e["config1"]["sampler_ms"], "at_scale", e["config1"]["at_scale"]["value"], e["config1"]["at_scale"]["units_per_s"], e["config1"]["at_scale"]["sampler_ms"])
print("dedup", e["config2_dedup"]["value"]); print("moving", e["moving"]["reads_16"]["value"], e["moving"]["reads_40"]["value"]); print("c4", e["config4"]["value"], e["config4"]["roofline"]["valu_issue_frac"], "c5", e["config5"]["value"], "e2e", e["program_e2e"]["value"])
print("cpu", j["cpu_baseline"]["value"])
PY
