cd /root/repo
mkdir -p gpurun_out/r03z
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_denovo.py -q -x -k "config5 or deep or shapes or read or 1000 or config2 or wide" > gpurun_out/r03z/pytest.txt 2>&1; tail -4 gpurun_out/r03z/pytest.txt
python - <<'PY' > gpurun_out/r03z/deep_fuzz.txt 2>&1
import sys
sys.path.insert(0, "/root/repo/tests")
import fuzz_kernels
bad = fuzz_kernels.run(12, 5001, ploidies=(4, 5, 6, 8), read_depths=(260, 400, 700, 1000), tempering=False, max_pos=20)
bad += fuzz_kernels.run(6, 5002, ploidies=(2, 3, 4), read_depths=(300, 600, 1500, 3000), tempering=False, max_pos=12)
print("DEEP FUZZ FAILURES", bad)
PY
tail -n 3 gpurun_out/r03z/deep_fuzz.txt
for L in 64 256; do python tools/config5_once.py $L 2 2>&1 | grep -v amdgpu | tail -2; done
MCHAP_HIP_FLAGS=512 python tools/config5_once.py 256 2 2>&1 | grep -v amdgpu | tail -1
