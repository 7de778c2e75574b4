import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from mchap_amd import DenovoMCMC
from mchap_amd.synth import synth_units
from mchap_amd.classes import sort_haplotypes
from test_gpu_denovo import _oracle_trace
K, M, A, R = 6, 4, 2, 70
reads, _, _ = synth_units(2, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=7, window=(2, M))
cases = {
    "all three p=1": dict(recombination_step_probability=1.0, partial_dosage_step_probability=1.0, dosage_step_probability=1.0),
    "default": dict(),
    "r.5 + f1": dict(recombination_step_probability=0.5, partial_dosage_step_probability=-1.0, dosage_step_probability=1.0),
    "p.5 + f1": dict(recombination_step_probability=-1.0, partial_dosage_step_probability=0.5, dosage_step_probability=1.0),
    "r.5 + p.5": dict(recombination_step_probability=0.5, partial_dosage_step_probability=0.5, dosage_step_probability=-1.0),
    "r1 + p.5 + f1": dict(recombination_step_probability=1.0, partial_dosage_step_probability=0.5, dosage_step_probability=1.0),
    "r.5 + p1 + f1": dict(recombination_step_probability=0.5, partial_dosage_step_probability=1.0, dosage_step_probability=1.0),
}
def firsts(tr, ref):
    out = []
    for u in range(2):
        for c in range(3):
            d = (tr[u].genotypes[c] != ref[u][c]).any(axis=(-1, -2))
            out.append(int(np.argmax(d)) if d.any() else None)
    return out
for name, extra in cases.items():
    kw = dict(ploidy=K, n_alleles=[A] * M, inbreeding=None, steps=200, chains=3, random_seed=11, **extra)
    m = DenovoMCMC(kernel=2, **kw)
    ref = [sort_haplotypes(_oracle_trace(m, reads[u], m.n_alleles, None, u)[0]) for u in range(2)]
    for k in (3,):
        tr = DenovoMCMC(kernel=k, **kw).fit_batch(list(reads))
        print("  %-16s kernel %d vs oracle: %s" % (name, k, firsts(tr, ref)), flush=True)
