cd /root/repo
mkdir -p gpurun_out/r03x
timeout 900 python tools/soak.py 3000 > gpurun_out/r03x/soak.txt 2>&1; tail -n 3 gpurun_out/r03x/soak.txt
timeout 700 python tests/fuzz_ragged.py 60 7 600 24 > gpurun_out/r03x/ragged7.txt 2>&1; tail -n 2 gpurun_out/r03x/ragged7.txt
timeout 700 python tests/fuzz_ragged.py 60 8 300 30 > gpurun_out/r03x/ragged8.txt 2>&1; tail -n 2 gpurun_out/r03x/ragged8.txt
grep -c " ok" gpurun_out/r03x/ragged7.txt gpurun_out/r03x/ragged8.txt; grep -h "FAIL" gpurun_out/r03x/*.txt | head
