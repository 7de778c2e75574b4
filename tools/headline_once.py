"""Measurement aid: the sampler call of BASELINE.json configs[1] (10 000 loci) one pass at a time: HIP-event span of the sampler's
launches under the environment's MCHAP_HIP_* knobs.  Usage: python tools/headline_once.py [loci] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reads, _, _ = synth_units(U)
b = DenovoDeviceBatch(DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42), reads)
b.time_sampler(True)
ms = []
for _ in range(reps):
    b.run()
    torch.cuda.synchronize()
    ms.append(b.sampler_ms())
print("%s: sampler %s ms (median %.3f)  %s" % (b.sampler_name, " ".join("%.2f" % m for m in ms), sorted(ms)[len(ms) // 2],
                                              {k: v for k, v in os.environ.items() if k.startswith("MCHAP_HIP_")}), flush=True)
