import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mchap_amd import DenovoMCMC, application, io
samples, targets, variants, matrices, contigs = application.load_matrices("tests/golden/example_biparental.npz")
source = application.MatrixSource(samples, matrices)
name, sample = sys.argv[1], sys.argv[2]
t = [x for x in targets if x[3] == name][0]
locus = io.DenovoLocus(t[0], t[1], t[2], t[3], variants, "N" * (t[2] - t[1]))
sr = source.reads(locus, sample)
def run(env):
    for k in list(os.environ):
        if k.startswith("MCHAP_HIP_"):
            del os.environ[k]
    os.environ.update(env)
    m = DenovoMCMC(ploidy=4, n_alleles=list(locus.n_alleles), steps=600, chains=2, random_seed=42)
    return m.fit_batch([sr["dists"]], [sr["counts"]], stream_ids=[0])[0]
ref = run({"MCHAP_HIP_KERNEL": "3"})
for env in ({"MCHAP_HIP_FLAGS": "64"}, {"MCHAP_HIP_FLAGS": "64", "MCHAP_HIP_ROUNDS": "6"}, {"MCHAP_HIP_FLAGS": "64", "MCHAP_HIP_PIPE_MAX": "16"}, {"MCHAP_HIP_FLAGS": "64", "MCHAP_HIP_PIPE_MAX": "200"}):
    tr = run(env)
    bad = np.flatnonzero(tr.llks[0] != ref.llks[0])
    badg = np.flatnonzero((tr.genotypes[0] != ref.genotypes[0]).any(axis=(1, 2)))
    print(env, "chain0 differing llk rows:", bad.tolist()[:80], "genotype rows:", badg.tolist()[:80])
    bad1 = np.flatnonzero(tr.llks[1] != ref.llks[1])
    print("   chain1 differing rows:", bad1.tolist()[:40])
