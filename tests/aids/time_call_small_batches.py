import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from mchap_amd.calling_mcmc import CallingMCMC
from mchap_amd.synth import synth_units
for (U, K, H, M, R) in ((150, 4, 6, 8, 60), (150, 4, 3, 8, 60), (600, 4, 6, 8, 60)):
    rng = np.random.default_rng(1)
    reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=5)
    haps = np.zeros((U, H, M), np.int8)
    for u in range(U):
        pool = np.unique(np.concatenate([truth[u], rng.integers(0, 2, size=(6 * H, M)).astype(np.int8)]), axis=0)
        rng.shuffle(pool); haps[u] = pool[:H]
    model = CallingMCMC(ploidy=K, haplotypes=haps[0], steps=2000, chains=2, random_seed=3)
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        s = model.fit_batch_summaries(reads, None, haplotypes=haps, prior=(np.full(U, 0.1), None), burn=1000)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("U %d K %d H %d: fit_batch_summaries %.1f ms" % (U, K, H, dt * 1e3), flush=True)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        s = model.fit_batch(reads, None, haplotypes=haps, prior=(np.full(U, 0.1), None))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("   fit_batch (host traces) %.1f ms" % (dt * 1e3), flush=True)
