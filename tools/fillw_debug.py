"""Debug aid: one small phased-sampler batch under the MCHAP_HIP_FLAGS of the environment; prints the status counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units
U = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reads, _, _ = synth_units(U)
b = DenovoDeviceBatch(DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=100, chains=2, random_seed=42, kernel=5), reads)
b.run()
torch.cuda.synchronize()
w, f, l, st = b.traces()
print("ok", os.environ.get("MCHAP_HIP_FLAGS"), np.unique(st), float(l.sum()))
