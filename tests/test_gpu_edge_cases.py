"""GPU: the edge cases the reference's own operator tests hold (tests/test_assemble/test_mcmc.py:95-247), with the
reference's assertions, through mchap_amd.DenovoMCMC."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _model(ploidy, n_base, steps=1000, chains=2, **kw):
    from mchap_amd import DenovoMCMC

    return DenovoMCMC(ploidy=ploidy, n_alleles=np.array([2] * n_base, dtype=int), steps=steps, chains=chains, random_seed=7, **kw)


def _reads_of(haplotypes, n_reads, seed, prob=0.999999):
    """Error-free reads (qual 60) sampled uniformly from the haplotypes, as mchap.testing.simulate_reads does."""
    rng = np.random.default_rng(seed)
    haplotypes = np.asarray(haplotypes)
    src = haplotypes[rng.integers(0, len(haplotypes), size=n_reads)]
    reads = np.full(src.shape + (2,), (1.0 - prob) / 3.0)  # error_factor 3 (encoding/integer/transcode.py:16)
    np.put_along_axis(reads, src[..., None], prob, axis=-1)
    return reads


def test_zero_reads():
    # test_DenovoMCMC__zero_reads (95-111)
    trace = _model(4, 6).fit(np.empty((0, 6, 2), dtype=float)).burn(500)
    assert trace.genotypes.shape == (2, 500, 4, 6)
    assert trace.posterior().probabilities[0] < 0.05


@pytest.mark.parametrize("n_reads", [10, 0])
def test_zero_snps(n_reads):
    # test_DenovoMCMC__zero_snps / __zero_reads_or_snps (114-149)
    from mchap_amd import DenovoMCMC

    model = DenovoMCMC(ploidy=4, n_alleles=np.array([], int), steps=1000, chains=2, random_seed=7)
    trace = model.fit(np.empty((n_reads, 0, 0), dtype=float)).burn(500)
    assert trace.genotypes.shape == (2, 500, 4, 0)
    assert np.all(np.isnan(trace.llks))
    assert trace.posterior().probabilities[0] == 1


def test_all_nans():
    # test_DenovoMCMC__all_nans (152-169)
    reads = np.full((10, 6, 2), np.nan)
    trace = _model(4, 6).fit(reads).burn(500)
    assert trace.genotypes.shape == (2, 500, 4, 6)
    assert trace.posterior().probabilities[0] < 0.05


def test_nan_reads_are_ignored():
    # test_DenovoMCMC__nans_ignored (172-211): all-NaN reads must not change the trace
    haplotypes = np.array([[0, 0, 0, 0, 0, 0], [0, 1, 0, 1, 1, 1], [0, 1, 0, 1, 1, 1], [1, 1, 1, 1, 1, 1]])
    reads1 = _reads_of(haplotypes, 16, 1)
    reads2 = np.concatenate([reads1, np.full((4, 6, 2), np.nan)])
    model = _model(4, 6)
    t1, t2 = model.fit(reads1).burn(500), model.fit(reads2).burn(500)
    np.testing.assert_array_equal(t1.genotypes, t2.genotypes)


def test_non_variable_locus():
    # test_DenovoMCMC__non_variable (214-247): every position fixed homozygous -> constant trace, nan llks
    haplotypes = np.array([[0, 0, 0, 1, 1, 1]] * 3)
    reads = _reads_of(haplotypes, 40, 2)
    trace = _model(3, 6, steps=100).fit(reads).burn(50)
    posterior = trace.posterior()
    assert trace.genotypes.shape == (2, 50, 3, 6)
    assert np.isnan(trace.llks).all()
    assert len(posterior.genotypes) == 1
    np.testing.assert_array_equal(haplotypes, posterior.genotypes[0])
    assert posterior.probabilities[0] == 1.0


def test_seed_reproducibility_and_difference():
    # test_DenovoMCMC__seed (354-397): same seed -> same trace, another seed -> another trace
    from mchap_amd import DenovoMCMC

    haplotypes = np.array([[0, 0, 0, 0, 0, 0], [0, 1, 0, 1, 1, 1], [0, 1, 0, 1, 1, 1], [1, 1, 1, 1, 1, 1]])
    reads = _reads_of(haplotypes, 8, 3, prob=0.9)
    kw = dict(ploidy=4, n_alleles=[2] * 6, steps=300, chains=2)
    a = DenovoMCMC(random_seed=42, **kw).fit(reads)
    b = DenovoMCMC(random_seed=42, **kw).fit(reads)
    c = DenovoMCMC(random_seed=36, **kw).fit(reads)
    np.testing.assert_array_equal(a.genotypes, b.genotypes)
    np.testing.assert_array_equal(a.llks, b.llks)
    assert not np.array_equal(a.genotypes, c.genotypes)


@pytest.mark.parametrize("shape", [(4, 13, 2, None), (4, 23, 2, None), (2, 9, 2, 0.2), (6, 10, 2, None), (3, 7, 3, 0.1), (8, 12, 2, None)],
                         ids=lambda s: "K%d-M%d-A%d-F%s" % s)
def test_samples_without_reads_equal_the_oracle_on_every_kernel(shape, monkeypatch):
    """A sample without reads at a locus is sampled like any other (one all-gap read: assemble/mcmc.py:132-137), and so is one
    whose reads are gaps at every SNV.  Every genotype then has the same likelihood, which the fast samplers use (no cache probe, no
    evaluation: denovo_spec_kernel's spec_eval) -- these chains move at every other sub-step and were the slowest of a launch.
    Traces against the oracle, step for step, on the default dispatch, the speculative kernel and the lanes-over-chains kernel
    (which has no shortcut), with the shortcut switched off (tuning flag 128) as well; a unit WITH reads shares the launch."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units
    from oracle import binding as orc
    from tests.helpers import beta_break_table

    K, M, A, F = shape
    steps = 150
    n_alleles = [A] * M
    if A > 2:
        n_alleles[1] = 2  # a biallelic position in a triallelic unit: its third row is 0.0 and never read
    gaps1 = np.full((1, M, A), np.nan)
    gaps5 = np.full((5, M, A), np.nan)
    for rd in (gaps1, gaps5):
        for j, n in enumerate(n_alleles):
            rd[:, j, n:] = 0.0
    real, _, _ = synth_units(1, ploidy=K, n_pos=M, n_reads=40, n_alleles=A, first_unit=900, window=(max(1, M // 2), M))
    for j, n in enumerate(n_alleles):
        real[0][:, j, n:] = 0.0
    reads = [gaps1, real[0], gaps5]
    ref = []
    for u, rd in enumerate(reads):
        cfg = orc.make_cfg(K, steps, 2, F, (1.0,), llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX, seed=3, stream_id=u,
                           break_table=beta_break_table(M, 1.0, 3.0))
        g, l, code = orc.denovo_fit(cfg, rd, n_alleles)
        assert code == 0
        ref.append((sort_haplotypes(g), l))
    assert (np.diff(ref[0][1], axis=1) == 0).all() and (np.diff(ref[0][0], axis=1) != 0).any()  # constant llk, moving chain
    for kernel, flags in ((0, 0), (0, 128), (0, 256), (3, 0), (2, 0)):  # (256: the instantiation without side-by-side evaluation)
        monkeypatch.setenv("MCHAP_HIP_KERNEL", str(kernel))
        monkeypatch.setenv("MCHAP_HIP_FLAGS", str(flags))
        model = DenovoMCMC(ploidy=K, n_alleles=n_alleles, inbreeding=F, steps=steps, chains=2, random_seed=3)
        traces = model.fit_batch(reads)
        for u, tr in enumerate(traces):
            assert np.array_equal(tr.genotypes, ref[u][0]), "kernel %d flags %d unit %d" % (kernel, flags, u)
            np.testing.assert_allclose(tr.llks, ref[u][1], rtol=1e-10, atol=1e-12)
