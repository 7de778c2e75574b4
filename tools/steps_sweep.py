"""Measurement aid (not part of the product): sampler kernel time against the number of MCMC steps and the batch size,
to separate the start-up transient (memo tables filling, chains converging) from the steady state.
    python tools/steps_sweep.py [loci]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reads, _, _ = synth_units(U)
for steps in (10, 25, 50, 100, 200, 400, 1000):
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=steps, chains=2, random_seed=42)
    b = DenovoDeviceBatch(model, reads)
    b.time_sampler(True)
    ms = []
    for _ in range(3):
        b.run()
        torch.cuda.synchronize()
        ms.append(b.sampler_ms())
    print("loci %d steps %4d  kernel %s  ms %s" % (U, steps, b.sampler_name, " ".join("%.3f" % m for m in ms)), flush=True)
    del b
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
for n in (2048, 4096, 8192, 10000, 12288, 16384, 20000, 40000):
    if n > U:
        r2, _, _ = synth_units(n)
    else:
        r2 = reads[:n]
    b = DenovoDeviceBatch(model, r2)
    b.time_sampler(True)
    ms = []
    for _ in range(2):
        b.run()
        torch.cuda.synchronize()
        ms.append(b.sampler_ms())
    print("loci %5d steps 1000 ms %s -> %.0f loci/s (sampler only)" % (n, " ".join("%.3f" % m for m in ms), n / (ms[-1] * 1e-3)), flush=True)
    del b
