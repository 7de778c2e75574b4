cd /root/repo
mkdir -p gpurun_out/r03m
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -x -k "wide or config5 or config2" > gpurun_out/r03m/pytest_wide.txt 2>&1
tail -15 gpurun_out/r03m/pytest_wide.txt
python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03m/bench.json 2> gpurun_out/r03m/bench.err
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/r03m/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], "one", j.get("value_one_in_flight"), "kernel_ms", j["roofline"]["kernel_ms"])
PY
export MCHAP_BENCH_DIST=1
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r03m/rccl1.json 2> gpurun_out/r03m/rccl1.err; echo rc=$?; tail -c 600 gpurun_out/r03m/rccl1.json; tail -5 gpurun_out/r03m/rccl1.err
