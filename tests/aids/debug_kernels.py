"""Development aid: first step at which a sampler kernel's trace leaves the oracle's, for one of a few named cases.
    python tests/debug_kernels.py <kernel> <case> [flags]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from mchap_amd import DenovoMCMC
from mchap_amd.classes import sort_haplotypes
from mchap_amd.synth import synth_units
from test_gpu_denovo import _oracle_trace

kernel = int(sys.argv[1]); case = sys.argv[2]
if len(sys.argv) > 3:
    os.environ["MCHAP_HIP_FLAGS"] = sys.argv[3]
if case == "a":
    reads, _, _ = synth_units(3, ploidy=4, n_pos=6, n_reads=48, first_unit=100)
    kw = dict(ploidy=4, n_alleles=[2] * 6, inbreeding=None, steps=120, chains=2, random_seed=7)
elif case == "b":
    reads, _, _ = synth_units(6, ploidy=4, n_pos=8, n_reads=200)
    kw = dict(ploidy=4, n_alleles=[2] * 8, steps=150, chains=2, random_seed=42)
elif case == "c":
    reads, _, _ = synth_units(2, ploidy=6, n_pos=4, n_reads=70, n_alleles=2, first_unit=7, window=(2, 4))
    kw = dict(ploidy=6, n_alleles=[2] * 4, inbreeding=0.05, steps=60, chains=3, random_seed=11)
tr = DenovoMCMC(kernel=kernel, **kw).fit_batch(list(reads))
m0 = DenovoMCMC(kernel=2, **kw)
for u in range(len(reads)):
    g, l = _oracle_trace(m0, reads[u], m0.n_alleles, None, u)
    g = sort_haplotypes(g)
    for c in range(g.shape[0]):
        bad = [s for s in range(g.shape[1]) if not np.array_equal(g[c, s], tr[u].genotypes[c, s]) or not np.isclose(l[c, s], tr[u].llks[c, s], rtol=1e-10)]
        chg = [s for s in range(1, g.shape[1]) if not np.array_equal(g[c, s], g[c, s - 1])]
        print("unit", u, "chain", c, "first mismatch", bad[:3], "n bad", len(bad), "| oracle genotype changes at", chg[:12])
        if bad:
            s = bad[0]
            print("  oracle:", g[c, s].tolist(), l[c, s], "\n  kernel:", tr[u].genotypes[c, s].tolist(), tr[u].llks[c, s])
