"""Profiling helper (not part of the product): one sampler launch for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

mode = sys.argv[1] if len(sys.argv) > 1 else "full"
U = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
kw = {}
if mode == "nostruct":
    kw = dict(recombination_step_probability=-1.0, partial_dosage_step_probability=-1.0, dosage_step_probability=-1.0)
reads, _, _ = synth_units(U)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=steps, chains=2, random_seed=42, **kw)
b = DenovoDeviceBatch(model, reads)
b.run()
b.posterior(steps // 2)
torch.cuda.synchronize()
print("done", mode, U, steps)
