"""debug aid: one docs/example unit through kernel 5 with several hand-over schedules against kernel 3"""
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
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 600
def run(env):
    for k in list(os.environ):
        if k.startswith("MCHAP_HIP_"):
            del os.environ[k]
    os.environ.update(env)
    m = DenovoMCMC(ploidy=4, n_alleles=list(locus.n_alleles), steps=STEPS, chains=2, random_seed=42)
    tr = m.fit_batch([sr["dists"]], [sr["counts"]], stream_ids=[0])[0]
    return tr, m.last_sampler
ref, nm = run({"MCHAP_HIP_KERNEL": "3"})
print("reads", sr["dists"].shape, nm)
for env in ({}, {"MCHAP_HIP_FLAGS": "64"}, {"MCHAP_HIP_ROUNDS": "0"}, {"MCHAP_HIP_PIPE_FIRST": "1"}, {"MCHAP_HIP_PIPE_FIRST": "32"}, {"MCHAP_HIP_PIPE_FIRST": "200"},
            {"MCHAP_HIP_PIPE_RESUME": "1"}, {"MCHAP_HIP_PIPE_RESUME": "64"}, {"MCHAP_HIP_ROUNDS": "6"}, {"MCHAP_HIP_PIPE_PARTS": "1"}, {"MCHAP_HIP_NO_BP_CACHE": "1"},
            {"MCHAP_HIP_CACHE_SLOTS": "64"}, {"MCHAP_HIP_KERNEL": "5", "MCHAP_HIP_TEST_KERNELS": "1", "MCHAP_HIP_PIPE_GROUP": "16"}):
    tr, nm = run(env)
    d = np.argwhere((tr.genotypes != ref.genotypes).any(axis=(2, 3)))
    dl = np.argwhere(tr.llks != ref.llks)
    print(env, nm, "first differing (chain, step):", d[0].tolist() if len(d) else None, "llk:", dl[0].tolist() if len(dl) else None, flush=True)
    if len(d):
        c, s0 = d[0]
        print("   ref llk", ref.llks[c, s0 - 2:s0 + 3], "\n   got llk", tr.llks[c, s0 - 2:s0 + 3])
        print("   moves in ref around:", np.flatnonzero(np.diff(ref.llks[c]) != 0)[:40].tolist())
