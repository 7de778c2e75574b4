"""Debug aid: the in-flight test's scenario, repeated; prints which batch / unit / chain / step differs from the sequential run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch, PassesInFlight
from mchap_amd.synth import synth_units

model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=200, chains=2, random_seed=7)
U = 300
inputs = [synth_units(U, first_unit=1000 * i)[0] for i in range(5)]
ref = []
for i, reads in enumerate(inputs):
    b = DenovoDeviceBatch(model, reads, first_stream=1000 * i)
    b.run(); b.posterior(100); torch.cuda.synchronize()
    ref.append(b.d_trace.cpu().numpy().copy().reshape(U, 2, 200, 4))
    # sequential repeat: deterministic?
    b.run(); torch.cuda.synchronize()
    again = b.d_trace.cpu().numpy().reshape(U, 2, 200, 4)
    if not np.array_equal(again, ref[-1]):
        d = np.argwhere((again != ref[-1]).any(axis=3))
        print("SEQUENTIAL repeat differs: batch", i, "first", d[0], "n", len(d))
    del b
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    batches = [DenovoDeviceBatch(model, reads, first_stream=1000 * i) for i, reads in enumerate(inputs)]
    flight = PassesInFlight(3)
    for rep in range(2):
        for b in batches[rep:] + batches[:rep]:
            flight.submit(lambda b=b: (b.run(), b.posterior(100)))
    flight.join(); torch.cuda.synchronize()
    for i, b in enumerate(batches):
        got = b.d_trace.cpu().numpy().reshape(U, 2, 200, 4)
        if not np.array_equal(got, ref[i]):
            d = np.argwhere((got != ref[i]).any(axis=3))
            print("trial", trial, "batch", i, "differs: units", sorted(set(d[:, 0].tolist()))[:10], "first (unit, chain, step)", d[0].tolist(), "n", len(d))
    print("trial", trial, "done", flush=True)
