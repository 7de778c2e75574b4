"""GPU, statistical: the sampler's stationary distribution against exact enumeration — the reference's own notion of
sampler correctness (tests/test_assemble/test_mutation.py:188-276, test_structural.py:373-672, test_mcmc.py:423-494:
long runs compared with the exact posterior to two decimals).  Here both sides run on the GPU: DenovoMCMC's trace
against mchap_exact_genotype_likelihoods / _posteriors over ALL haplotypes, with the priors the reference shows to
be equal (assemble.prior.log_genotype_prior == calling.prior.log_genotype_prior with flat frequencies,
tests/test_calling/test_calling_exact.py:91-164)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ploidy,n_pos,inbreeding,temps", [
    (2, 3, 0.0, (1.0,)),
    (4, 2, 0.0, (1.0,)),
    (4, 3, 0.2, (1.0,)),
    (2, 3, 0.1, (0.2, 0.6, 1.0)),
])
def test_posterior_matches_exact_enumeration(ploidy, n_pos, inbreeding, temps):
    from mchap_amd import DenovoMCMC, calling
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(1, ploidy=ploidy, n_pos=n_pos, n_reads=12, window=(2, n_pos), qual=(3, 12), first_unit=5)
    reads = reads[0]
    haps = np.array(list(itertools.product([0, 1], repeat=n_pos)), dtype=np.int8)  # every haplotype, lexicographic
    H = len(haps)
    llks = calling.genotype_likelihoods(reads, ploidy, haps).astype(np.float64)
    exact = calling.genotype_posteriors(llks, ploidy, H, prior=(inbreeding, None))
    assert abs(exact.sum() - 1.0) < 1e-9

    steps, burn = 30000, 1000
    model = DenovoMCMC(ploidy=ploidy, n_alleles=[2] * n_pos, inbreeding=inbreeding, steps=steps, chains=2, temperatures=temps,
                       fix_homozygous=1.1, random_seed=123)  # 1.1: never fix a position, every haplotype stays possible
    trace = model.fit(reads)
    g = trace.genotypes[:, burn:]                                   # [C, S, K, M], haplotypes sorted within a genotype
    weights = 2 ** np.arange(n_pos - 1, -1, -1)
    hidx = (g.astype(np.int64) * weights).sum(axis=-1)              # haplotype -> its index in `haps`
    hidx.sort(axis=-1)
    gidx = np.array([calling.genotype_alleles_as_index(a) for a in np.unique(hidx.reshape(-1, ploidy), axis=0)])
    uniq, counts = np.unique(hidx.reshape(-1, ploidy), axis=0, return_counts=True)
    sampled = np.zeros_like(exact)
    sampled[gidx] = counts / counts.sum()
    # the reference asserts two decimals on 25k-50k steps; 58k correlated samples here
    assert np.abs(sampled - exact).max() < 0.03, (np.abs(sampled - exact).max(), exact.max())
    assert np.argmax(sampled) == np.argmax(exact) or exact.max() - np.sort(exact)[-2] < 0.05
