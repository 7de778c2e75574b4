"""Measurement aid: the extra.call_mcmc workload of bench.py (`mchap call` sampler, 4096 tetraploid units) launched once or a few
times, for rocprofv3 --kernel-trace / --pmc passes (tools/profile_round.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

U = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
once, shapes, arrays = bench.call_mcmc_workload(U)
import time
for _ in range(reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    once()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("call sampler: %.1f ms -> %.0f units/s" % (dt * 1e3, U / dt), flush=True)
print("done", U, shapes)
