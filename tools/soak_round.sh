# Run ON THE GPU BOX via gpurun: differential soak (all kernels against the oracle, ragged launches, deep reads) -> gpurun_out/soak/.
cd /root/repo
mkdir -p gpurun_out/soak
timeout 900 python tools/soak.py 4000 > gpurun_out/soak/soak.txt 2>&1; tail -n 2 gpurun_out/soak/soak.txt
timeout 700 python tests/fuzz_ragged.py 50 11 500 24 > gpurun_out/soak/ragged11.txt 2>&1; tail -n 1 gpurun_out/soak/ragged11.txt
timeout 700 python tests/fuzz_ragged.py 50 12 200 40 > gpurun_out/soak/ragged12.txt 2>&1; tail -n 1 gpurun_out/soak/ragged12.txt
python - <<'PY' > gpurun_out/soak/deep_fuzz.txt 2>&1
import sys
sys.path.insert(0, "/root/repo/tests")
import fuzz_kernels
bad = fuzz_kernels.run(16, 6001, ploidies=(3, 4, 6, 8), read_depths=(257, 300, 520, 800, 1100), tempering=None, max_pos=22)
print("DEEP FUZZ FAILURES", bad)
PY
tail -n 2 gpurun_out/soak/deep_fuzz.txt
grep -c " ok" gpurun_out/soak/*.txt; grep -h "FAIL" gpurun_out/soak/*.txt | head
