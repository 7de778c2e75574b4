"""Measurement aid: sampler time of the steady-state pipeline against its number of settle/steady rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MCHAP_HIP_KERNEL", "4")
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units
U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = _lib.lib(); L.mchap_set_profiling(1)
reads, _, _ = synth_units(U)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
b = DenovoDeviceBatch(model, reads)
for rounds in (1, 2, 3, 4, 6, 8, 12):
    os.environ["MCHAP_HIP_ROUNDS"] = str(rounds)
    ms = []
    for _ in range(3):
        b.run(); torch.cuda.synchronize(); ms.append(L.mchap_last_sampler_ms())
    print("rounds %2d: %s ms" % (rounds, " ".join("%.2f" % m for m in ms)), flush=True)
