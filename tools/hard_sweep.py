"""Measurement aid: the phased sampler against kernel 3 on loci whose chains keep moving (shallow, low-quality reads),
i.e. where coasting cannot help: same traces required, sampler time of each.
    python tools/hard_sweep.py [loci] [reads]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cache = int(sys.argv[3]) if len(sys.argv) > 3 else 100  # llk_cache_threshold (-1: no cache)
reads, _, _ = synth_units(U, n_reads=R, qual=(3, 20))
out = {}
kernels = tuple(int(k) for k in os.environ.get("SWEEP_KERNELS", "3,0").split(","))
for kernel in kernels:
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42, kernel=kernel, llk_cache_threshold=cache)
    b = DenovoDeviceBatch(model, reads)
    b.time_sampler(True)
    ms = []
    for _ in range(2):
        b.run()
        torch.cuda.synchronize()
        ms.append(b.sampler_ms())
    tr = b.d_trace.cpu().numpy().reshape(U, 2, 1000, 4)
    out[kernel] = tr
    moved = (np.diff(tr[:, :, 100:], axis=2) != 0).any(axis=(2, 3)).mean()
    print("%d loci x %d reads  %-62s %s ms   chains moving after step 100: %.1f %%" % (
        U, R, b.sampler_name, " ".join("%.2f" % m for m in ms), 100 * moved), flush=True)
    del b
print("same traces" if all(np.array_equal(out[kernels[0]], out[k]) for k in kernels[1:]) else "TRACES DIFFER")
