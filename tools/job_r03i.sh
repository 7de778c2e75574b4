cd /root/repo
mkdir -p gpurun_out/r03i
export GPU_MAX_HW_QUEUES=16
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_exact.py tests/test_gpu_exact_properties.py tests/test_gpu_call_mcmc.py tests/test_gpu_call_exact_goldens.py tests/test_gpu_posterior.py -q -x > gpurun_out/r03i/pytest.txt 2>&1
tail -5 gpurun_out/r03i/pytest.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03i/c4_trace -- python3 /root/repo/tools/exact_once.py 256 streaming > /root/repo/gpurun_out/r03i/c4_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d /root/repo/gpurun_out/r03i/c4_sq1 -- python3 /root/repo/tools/exact_once.py 256 streaming > /root/repo/gpurun_out/r03i/c4_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r03i/c4_sq2 -- python3 /root/repo/tools/exact_once.py 256 streaming > /root/repo/gpurun_out/r03i/c4_sq2.log 2>&1
cd /root/repo
python3 tools/pmc_sq.py gpurun_out/r03i/c4_sq1 gpurun_out/r03i/c4_sq2 > gpurun_out/r03i/c4_sq.json
cat gpurun_out/r03i/c4_trace/*/*kernel_stats.csv | cut -c1-160
python bench.py --no-cpu-baseline > gpurun_out/r03i/bench.json 2> gpurun_out/r03i/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03i/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
e=j["extra"]
print("c4", json.dumps(e["config4"]))
PY
