"""Measurement aid: sampler time of 10 000 loci (8 SNVs, 200 reads, 1000 steps x 2 chains) at a given ploidy, for the
kernels / group sizes named by the environment.
    python tools/ploidy_sweep.py <ploidy> [loci]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

K = int(sys.argv[1])
U = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reads, _, _ = synth_units(U, ploidy=K)
model = DenovoMCMC(ploidy=K, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
b = DenovoDeviceBatch(model, reads)
b.time_sampler(True)
ms = []
for _ in range(3):
    b.run()
    torch.cuda.synchronize()
    ms.append(b.sampler_ms())
print("K=%d loci %d  %-64s %s  -> %.0f k loci/s (sampler only)  %s" % (K, U, b.sampler_name, " ".join("%.2f" % m for m in ms), U / min(ms),
      {k: v for k, v in os.environ.items() if k.startswith("MCHAP_HIP_")}), flush=True)
