"""Measurement aid: configs[4]'s tempered variant (one chain under the ladder 0.001 .. 1.0; octoploid, 20 SNVs, 1000 reads) --
replicas side by side, one wavefront each (default), or one after the other on one wavefront (MCHAP_HIP_FLAGS=16384).
    python tools/tempered_once.py [loci] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
reads, _, _ = synth_units(U, ploidy=8, n_pos=20, n_reads=1000, window=(8, 20), first_unit=77)
model = DenovoMCMC(ploidy=8, n_alleles=[2] * 20, steps=S, chains=1, temperatures=(0.001, 0.01, 0.1, 1.0), random_seed=42)
b = DenovoDeviceBatch(model, reads)
b.time_sampler(True)
t = time.perf_counter()
b.run()
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("%s: %d loci x %d steps: %.2f s (%.2f loci/s), sampler %.1f ms, flags %s" % (b.sampler_name, U, S, dt, U / dt, b.sampler_ms(), os.environ.get("MCHAP_HIP_FLAGS")))
