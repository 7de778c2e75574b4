cd /root/repo
mkdir -p gpurun_out/r03e
export GPU_MAX_HW_QUEUES=16
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03e/pytest.txt 2>&1
tail -6 gpurun_out/r03e/pytest.txt
python bench.py > gpurun_out/r03e/bench.json 2> gpurun_out/r03e/bench.err
tail -c 600 gpurun_out/r03e/bench.json; tail -3 gpurun_out/r03e/bench.err
