"""Measurement aid: the exact caller's streaming form at BASELINE configs[3] (256 units), ms per call; MCHAP_HIP_LIB picks the
library build (variants of pass 1's block size / occupancy); the mode of unit 0 is printed as a sanity check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import _lib
from mchap_amd.device import ExactDeviceBatch
from mchap_amd.synth import synth_units

U, K, H, M, R = 256, 6, 16, 10, 500
rng = np.random.default_rng(4)
reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(5, 10), first_unit=400)
haps = np.unique(rng.integers(0, 2, size=(64, M)).astype(np.int8), axis=0)[:H]
batch = ExactDeviceBatch(reads, K, haps, None, (0.1, rng.dirichlet(np.ones(H))))
for name, kw in (("streaming", dict(streaming=True, arrays=False)), ("arrays", dict(streaming=False, arrays=True))):
    batch.run(**kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        batch.run(**kw)
    e1.record()
    torch.cuda.synchronize()
    extra = batch.mode_results() if name == "streaming" else None
    print(os.path.basename(_lib.SO), name, "%.3f ms per call" % (e0.elapsed_time(e1) / 5),
          ("mode[0] %s llk %.12g gpm %.12g" % (extra[0][0].tolist(), extra[1][0], extra[2][0])) if extra else "")
