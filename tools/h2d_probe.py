"""Measurement aid: where the host time of a PCIe-inclusive pass goes (bench.py incl_h2d) with passes in flight."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import DenovoMCMC
from mchap_amd.device import DenovoDeviceBatch, PassesInFlight, pinned
from mchap_amd.synth import synth_units

U = 10000
_, calls, _ = synth_units(U)
quals = np.random.default_rng(5).integers(20, 41, size=calls.shape).astype(np.int16)
calls, quals = pinned(calls), pinned(quals)
model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
nfl = 4
flight = PassesInFlight(nfl)
slots = [None] * nfl
T = {"ctor": 0.0, "run": 0.0, "post": 0.0, "host": 0.0}
def turn(i, n, download=True):
    k = i % nfl
    with torch.cuda.stream(flight.streams[k]):
        if slots[k] is not None:
            t = time.perf_counter()
            if download:
                slots[k].posterior_host()
            else:
                flight.streams[k].synchronize()
            T["host"] += time.perf_counter() - t
            slots[k] = None
        if i < n:
            t = time.perf_counter(); b = DenovoDeviceBatch(model, None, calls=calls, quals=quals); T["ctor"] += time.perf_counter() - t
            t = time.perf_counter(); b.run(); T["run"] += time.perf_counter() - t
            t = time.perf_counter(); b.posterior(500); T["post"] += time.perf_counter() - t
            slots[k] = b
for download in (True, False):
    for i in range(2 * nfl):
        turn(i, nfl, download)
    torch.cuda.synchronize()
    for k in T: T[k] = 0.0
    n = 24
    t0 = time.perf_counter()
    for i in range(n + nfl):
        turn(i, n, download)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("download=%s: %.2f ms per pass; host ms per pass: %s" % (download, dt * 1e3, {k: round(v / n * 1e3, 2) for k, v in T.items()}), flush=True)
