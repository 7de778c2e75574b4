"""Profiling helper: one call of the exact caller at BASELINE configs[3] (hexaploid, 16 haplotypes x 10 SNVs, 500 reads) for
rocprofv3 --pmc / --kernel-trace runs.    python tools/exact_once.py [units] [streaming|arrays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd.device import ExactDeviceBatch
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "streaming"
K, H, M, R = 6, 16, 10, 500
rng = np.random.default_rng(4)
reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(5, 10), first_unit=400)
haps = np.unique(rng.integers(0, 2, size=(64, M)).astype(np.int8), axis=0)[:H]
batch = ExactDeviceBatch(reads, K, haps, None, (0.1, rng.dirichlet(np.ones(H))))
batch.run(streaming=mode == "streaming", arrays=mode != "streaming")
torch.cuda.synchronize()
print("done", U, mode, batch.G)
