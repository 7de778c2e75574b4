"""Measurement aid: one pass of BASELINE.json configs[4] (octoploid, 20 SNVs, 1000 reads, 4 chains x 2000 steps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reads, _, _ = synth_units(U, ploidy=8, n_pos=20, n_reads=1000, window=(8, 20), first_unit=77)
cache = int(sys.argv[3]) if len(sys.argv) > 3 else 100
model = DenovoMCMC(ploidy=8, n_alleles=[2] * 20, steps=2000, chains=4, random_seed=42, llk_cache_threshold=cache)
b = DenovoDeviceBatch(model, reads)
b.time_sampler(True)
for _ in range(reps):
    t = time.perf_counter()
    b.run()
    torch.cuda.synchronize()
    print("%s  %.1f ms sampler, %.1f ms wall -> %.1f loci/s  %s" % (b.sampler_name, b.sampler_ms(), (time.perf_counter() - t) * 1e3, U / (time.perf_counter() - t),
                                                         {k: v for k, v in os.environ.items() if k.startswith("MCHAP_HIP_")}), flush=True)
