"""Measurement aid: bench.py's extra.moving (chains that never settle) alone.  Usage: python tools/moving_once.py [reads] [loci] [cache_threshold]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
U = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cache = int(sys.argv[3]) if len(sys.argv) > 3 else 100
reads, _, _ = synth_units(U, ploidy=4, n_pos=8, n_reads=R, qual=(3, 20))
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42, llk_cache_threshold=cache)
b = DenovoDeviceBatch(model, reads)
b.time_sampler(True)
for _ in range(3):
    t = time.perf_counter()
    b.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("%d reads, %d loci, cache %d: %s  %.1f ms sampler -> %.0f loci/s  %s" % (R, U, cache, b.sampler_name, b.sampler_ms(), U / dt,
          {k: v for k, v in os.environ.items() if k.startswith("MCHAP_HIP_")}), flush=True)
