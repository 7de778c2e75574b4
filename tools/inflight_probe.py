"""Measurement aid: whole passes (prepare + sampler + posterior summary) of the headline workload with 1, 2 or 3
batches in flight on separate HIP streams.    python tools/inflight_probe.py [loci] [passes] [in-flight counts, e.g. 1,4]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import DenovoMCMC, _lib
from mchap_amd.device import DenovoDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reads, _, _ = synth_units(U)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
ref = None
NFL = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 3, 4, 6, 8]
for nfl in NFL:
    batches = [DenovoDeviceBatch(model, reads) for _ in range(nfl)]
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    def one(i):
        with torch.cuda.stream(streams[i % nfl]):
            batches[i % nfl].run()
            batches[i % nfl].posterior(500)
    for i in range(2 * nfl):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr = [b.d_trace.cpu().numpy() for b in batches]
    if ref is None:
        ref = tr[0].copy()
    same = all(np.array_equal(ref, t) for t in tr)
    print("%d in flight: %.2f ms per pass -> %.0f k loci/s   %s" % (nfl, dt / N * 1e3, U * N / dt / 1e3, "same traces" if same else "TRACES DIFFER"), flush=True)
    del batches
