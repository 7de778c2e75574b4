"""GPU: RNG-free properties of the exact caller that the reference's own tests assert
(tests/test_calling/test_calling_exact.py): enumeration == log_likelihood of the genotype (16-88), the streaming
posterior_mode == the full-array path (281-342, five decimals), flat frequencies == no frequencies (345-401)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [(2, 6, 4, 30), (4, 5, 6, 60), (4, 12, 8, 100), (6, 6, 5, 40), (3, 9, 7, 25)]  # ploidy, haplotypes, positions, reads


def _unit(K, H, M, R, seed):
    from mchap_amd.synth import synth_units

    rng = np.random.default_rng(seed)
    reads, _, _ = synth_units(1, ploidy=K, n_pos=M, n_reads=R, window=(max(1, M // 2), M), qual=(5, 40), first_unit=seed)
    haps = np.unique(rng.integers(0, 2, size=(8 * H, M)).astype(np.int8), axis=0)
    rng.shuffle(haps)
    counts = rng.integers(1, 4, size=R).astype(np.int64)
    return reads[0], haps[:H], counts


@pytest.mark.parametrize("K,H,M,R", SHAPES)
def test_enumeration_equals_log_likelihood(K, H, M, R):
    from mchap_amd import _lib, calling

    reads, haps, counts = _unit(K, H, M, R, 11)
    H = len(haps)
    llks = calling.genotype_likelihoods(reads, K, haps, read_counts=counts)
    G = calling.count_unique_genotypes(H, K)
    assert llks.shape == (G,) and llks.dtype == np.float32
    idx = np.unique(np.linspace(0, G - 1, 40).astype(int))
    genotypes = np.stack([haps[calling.index_as_genotype_alleles(int(i), K)] for i in idx]).astype(np.int8)
    out = np.zeros(len(idx))
    rd = np.ascontiguousarray(reads)
    _lib.check(_lib.lib().mchap_log_likelihood_batch(_lib.ptr(rd), R, M, 2, _lib.ptr(counts), _lib.ptr(genotypes), len(idx), K, _lib.ptr(out)))
    np.testing.assert_allclose(llks[idx], out.astype(np.float32), rtol=2e-6)  # float32 store (calling/exact.py:254)


@pytest.mark.parametrize("K,H,M,R", SHAPES)
@pytest.mark.parametrize("inbreeding", [0.0, 0.25])
def test_streaming_mode_equals_full_arrays(K, H, M, R, inbreeding):
    from mchap_amd import calling

    reads, haps, counts = _unit(K, H, M, R, 12)
    H = len(haps)
    freq = np.random.default_rng(5).dirichlet(np.ones(H))
    prior = (inbreeding, freq)
    alleles, mode_llk, mode_prob, support_prob, mean_freq, occur = calling.posterior_mode(
        reads, K, haps, read_counts=counts, prior=prior, return_support_prob=True, return_posterior_frequencies=True,
        return_posterior_occurrence=True)
    llks = calling.genotype_likelihoods(reads, K, haps, read_counts=counts).astype(np.float64)
    post = calling.genotype_posteriors(llks, K, H, prior=prior)
    i = int(np.argmax(post))
    np.testing.assert_array_equal(alleles, calling.index_as_genotype_alleles(i, K))
    np.testing.assert_almost_equal(mode_prob, post[i], decimal=5)
    np.testing.assert_almost_equal(mode_llk, llks[i], decimal=3)  # the array holds float32-rounded likelihoods
    f, _, o = calling.posterior_allele_frequencies(post, K, H)
    np.testing.assert_almost_equal(mean_freq, f, decimal=5)
    np.testing.assert_almost_equal(occur, o, decimal=5)
    alt = calling.alternate_dosage_posteriors(calling.index_as_genotype_alleles(i, K), post)
    np.testing.assert_almost_equal(support_prob, alt[1].sum(), decimal=5)


@pytest.mark.parametrize("K,H,M,R", SHAPES[:3])
def test_flat_frequencies_equal_no_frequencies(K, H, M, R):
    from mchap_amd import calling

    reads, haps, counts = _unit(K, H, M, R, 13)
    H = len(haps)
    llks = calling.genotype_likelihoods(reads, K, haps, read_counts=counts)
    a = calling.genotype_posteriors(llks, K, H, prior=(0.1, None))
    b = calling.genotype_posteriors(llks, K, H, prior=(0.1, np.full(H, 1.0 / H)))
    np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-9)  # float32 log-sum-exp on float32 likelihoods
    ma = calling.posterior_mode(reads, K, haps, read_counts=counts, prior=(0.1, None))
    mb = calling.posterior_mode(reads, K, haps, read_counts=counts, prior=(0.1, np.full(H, 1.0 / H)))
    np.testing.assert_array_equal(ma[0], mb[0])
    np.testing.assert_allclose(ma[1:], mb[1:], rtol=1e-9)
