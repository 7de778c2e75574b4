"""Profiling helper (not part of the product): wall time of the sampler under ablations."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reads, _, _ = synth_units(U)
base = dict(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
cases = {
    "full": {},
    "no structural": dict(recombination_step_probability=-1.0, partial_dosage_step_probability=-1.0, dosage_step_probability=-1.0),
    "only full dosage": dict(recombination_step_probability=-1.0, partial_dosage_step_probability=-1.0),
    "no cache": dict(llk_cache_threshold=-1),
    "kernel1 (wave/chain)": dict(kernel=1),
}
for name, kw in cases.items():
    model = DenovoMCMC(**base, **kw)
    b = DenovoDeviceBatch(model, reads)
    b.run(); torch.cuda.synchronize()
    t = time.time(); b.run(); torch.cuda.synchronize(); dt = time.time() - t
    print("%-24s %8.1f ms   (%d units)" % (name, dt * 1e3, U))
