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
STEPS = 600
def run(env):
    for k in list(os.environ):
        if k.startswith("MCHAP_HIP_"):
            del os.environ[k]
    os.environ.update(env)
    m = DenovoMCMC(ploidy=4, n_alleles=list(locus.n_alleles), steps=STEPS, chains=2, random_seed=42)
    return m.fit_batch([sr["dists"]], [sr["counts"]], stream_ids=[0])[0]
ref = run({"MCHAP_HIP_KERNEL": "3"})
mv = np.flatnonzero(np.diff(ref.llks[0]) != 0) + 1
print("chain 0 steps WITHOUT a move (first 120):", [s for s in range(1, 120) if s not in set(mv.tolist())])
for first in (1, 2, 3, 4, 5, 6, 8, 12, 16, 24):
    for rounds in (1, 2):
        for resume in (4, 8, 16):
            tr = run({"MCHAP_HIP_PIPE_FIRST": str(first), "MCHAP_HIP_ROUNDS": str(rounds), "MCHAP_HIP_PIPE_RESUME": str(resume), "MCHAP_HIP_FLAGS": "64"})
            d = np.argwhere((tr.genotypes != ref.genotypes).any(axis=(2, 3)))
            print("first", first, "rounds", rounds, "resume", resume, "->", d[0].tolist() if len(d) else "ok", flush=True)
