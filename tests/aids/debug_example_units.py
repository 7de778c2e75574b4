"""debug aid: docs/example units of one shape through several kernel settings against the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import numpy as np
from oracle import binding as orc
from tests.helpers import beta_break_table
from mchap_amd import DenovoMCMC, application, io
from mchap_amd.classes import sort_haplotypes

samples, targets, variants, matrices, contigs = application.load_matrices("tests/golden/example_biparental.npz")
source = application.MatrixSource(samples, matrices)
by = {}
for contig, start, stop, name in targets:
    locus = io.DenovoLocus(contig, start, stop, name, variants, "N" * (stop - start), sequence_known=False)
    for s in samples:
        sr = source.reads(locus, s)
        if len(sr["dists"]):
            by.setdefault(tuple(locus.n_alleles), []).append((sr["dists"], sr["counts"], name, s))
STEPS = 600
for n_alleles, units in sorted(by.items()):
    M = len(n_alleles)
    ref = []
    for rd, rc, name, s in units:
        cfg = orc.make_cfg(4, STEPS, 2, None, (1.0,), llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX, seed=42, stream_id=0, break_table=beta_break_table(M, 1.0, 3.0))
        g, l, code = orc.denovo_fit(cfg, rd, list(n_alleles), rc)
        ref.append(sort_haplotypes(g))
    for label, env in (("default", {}), ("old-fill", {"MCHAP_HIP_FLAGS": "64"}), ("kernel3", {"MCHAP_HIP_KERNEL": "3"}), ("kernel2", {"MCHAP_HIP_KERNEL": "2"})):
        for k in ("MCHAP_HIP_FLAGS", "MCHAP_HIP_KERNEL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        model = DenovoMCMC(ploidy=4, n_alleles=list(n_alleles), steps=STEPS, chains=2, random_seed=42)
        tr = model.fit_batch([u[0] for u in units], [u[1] for u in units], stream_ids=[0] * len(units))
        bad = [(i, units[i][2], units[i][3], len(units[i][0]), int(np.argwhere((tr[i].genotypes != ref[i]).any(axis=(2, 3)))[0][1])) for i in range(len(units)) if not np.array_equal(tr[i].genotypes, ref[i])]
        print("M", M, n_alleles if max(n_alleles) > 2 else "", label, model.last_sampler, "units", len(units), "bad", bad[:6], flush=True)
