# Run ON THE GPU BOX via gpurun from the repo root: smoke, the whole GPU suite, the default bench line (gpurun_out/final/).
cd /root/repo
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/final/pytest.txt 2>&1
tail -3 gpurun_out/final/pytest.txt
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/final/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "h2d", j["value_incl_h2d"]["value"])
e=j["extra"]
print("config1", e["config1"]["value"], e["config1"]["sampler_ms"], "at_scale", e["config1"]["at_scale"]["value"], e["config1"]["at_scale"]["units_per_s"])
print("dedup", e["config2_dedup"]["value"]); print("moving", e["moving"]["reads_16"]["value"], e["moving"]["reads_40"]["value"]); print("c4", e["config4"]["value"], "c5", e["config5"]["value"], e["config5"]["value_one_in_flight"], "e2e", e["program_e2e"]["value"])
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("mode_call_concordance"))
PY
