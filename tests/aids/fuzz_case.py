"""One explicit fuzz case for one kernel (development aid): python tests/fuzz_case.py kernel K M A R chains U T F steps lo seed cache p0 p1 p2 case"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from mchap_amd import DenovoMCMC
from mchap_amd.classes import sort_haplotypes
from mchap_amd.synth import synth_units
from test_gpu_denovo import _oracle_trace
a = sys.argv[1:]
k, K, M, A, R, chains, U, T = [int(x) for x in a[:8]]
F = None if a[8] == "None" else float(a[8])
steps, lo, seed, cache = [int(x) for x in a[9:13]]
pr = [float(x) for x in a[13:16]]
case = int(a[16])
temps = {1: (1.0,), 2: (0.3, 1.0), 3: (0.2, 0.6, 1.0)}[T]
reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=1000 + case, window=(lo, M), qual=(5, 40))
kw = dict(ploidy=K, n_alleles=[A] * M, inbreeding=F, steps=steps, chains=chains, temperatures=temps, random_seed=seed,
          recombination_step_probability=pr[0], partial_dosage_step_probability=pr[1], dosage_step_probability=pr[2],
          llk_cache_threshold=cache)
print("running kernel", k, flush=True)
tr = DenovoMCMC(kernel=k, **kw).fit_batch(list(reads))
m0 = DenovoMCMC(kernel=2, **kw)
ok = True
for u in range(U):
    g, l = _oracle_trace(m0, reads[u], m0.n_alleles, None, u)
    ok = ok and np.array_equal(tr[u].genotypes, sort_haplotypes(g))
print("kernel", k, "ok" if ok else "FAIL", flush=True)
