"""Measurement aid: one batch of the headline workload through the sampler named by MCHAP_HIP_KERNEL (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reads, _, _ = synth_units(U)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
b = DenovoDeviceBatch(model, reads)
for _ in range(reps):
    b.run()
    torch.cuda.synchronize()
