"""Measurement aid: DenovoMCMC.fit_batch through the host-pointer entry point (mchap_denovo_fit_batch allocates and frees its device
buffers per call) at small and medium batch sizes, under the environment's MCHAP_HIP_* knobs.  Usage: python tools/hostcall_once.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mchap_amd import DenovoMCMC
from mchap_amd.synth import synth_units

for U, K, M, R in ((8, 4, 8, 200), (100, 4, 8, 200), (400, 4, 8, 200), (1000, 4, 8, 200), (100, 4, 20, 60), (400, 4, 20, 60)):
    reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(4, M))
    model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=1000, chains=2, random_seed=42)
    rl = list(reads)
    model.fit_batch(rl)
    ts = []
    for _ in range(4):
        t = time.perf_counter()
        model.fit_batch(rl)
        ts.append((time.perf_counter() - t) * 1e3)
    print("%4d units K=%d M=%2d R=%3d: fit_batch %s ms  %s" % (U, K, M, R, " ".join("%.1f" % x for x in ts), {k: v for k, v in os.environ.items() if k.startswith("MCHAP_HIP_")}), flush=True)
