"""Measurement aid: the docs/example pileups (tests/golden/example_biparental.npz) one target at a time -- sampler ms of the
22 units of each target (2000 steps x 2 chains), its SNV count and read rows: which shapes set the 440-unit launch's time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mchap_amd import DenovoMCMC, application, io
from mchap_amd.device import DenovoRaggedBatch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
samples, targets, variants, matrices, contigs = application.load_matrices(os.path.join(root, "tests", "golden", "example_biparental.npz"))
source = application.MatrixSource(samples, matrices)
model = DenovoMCMC(ploidy=4, n_alleles=[2], steps=2000, chains=2, random_seed=42)
allu = []
for contig, start, stop, name in targets:
    locus = io.DenovoLocus(contig, start, stop, name, variants, "N" * (stop - start), sequence_known=False)
    M = len(locus.positions)
    units = []
    for s in samples:
        sr = source.reads(locus, s)
        if M and len(sr["dists"]):
            units.append(dict(reads=sr["dists"], counts=sr["counts"], n_alleles=locus.n_alleles, ploidy=4, inbreeding=None, stream_id=0))
    if not units:
        continue
    allu += units
    b = DenovoRaggedBatch(model, units)
    b.run(1000); torch.cuda.synchronize()
    t = time.perf_counter(); b.run(1000); torch.cuda.synchronize(); dt = time.perf_counter() - t
    tr = b.d_trace.cpu().numpy().view(np.uint64)
    moved = []
    for i, u in enumerate(units):
        D = b.units_host[i]
        w = tr[int(D["trace_off"]): int(D["trace_off"]) + 2 * 2000 * 4].reshape(2, 2000, 4)
        moved.append(int((np.diff(w, axis=1) != 0).any(axis=2).sum()))
    rows = [len(u["reads"]) for u in units]
    print("%s M=%2d units=%2d rows %3d..%3d  %.1f ms   moves per unit (of 4000 chain-steps): median %d max %d" % (
        name, M, len(units), min(rows), max(rows), dt * 1e3, int(np.median(moved)), max(moved)), flush=True)
b = DenovoRaggedBatch(model, allu)
b.run(1000); torch.cuda.synchronize()
t = time.perf_counter(); b.run(1000); torch.cuda.synchronize()
print("all %d units: %.1f ms" % (len(allu), (time.perf_counter() - t) * 1e3))
