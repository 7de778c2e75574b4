"""GPU parity of `mchap call`'s sampler (call_mcmc_kernel behind CallingMCMC) against the CPU oracle on the same Philox
streams -- alleles bit-exact step for step, llks to 1e-10 -- and, independently of any generator, its stationary
distribution against the exact caller's enumeration (the reference's own criterion,
tests/test_calling/test_calling_mcmc.py: Gibbs == Metropolis-Hastings == exact to two decimals)."""
import numpy as np
import pytest

from oracle import binding as orc

pytestmark = pytest.mark.gpu


def _inputs(U, K, H, M, R, seed, qual=(5, 25)):
    from mchap_amd.synth import synth_units

    rng = np.random.default_rng(seed)
    reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=seed, window=(2, M), qual=qual)
    haps = np.zeros((U, H, M), np.int8)
    for u in range(U):
        pool = np.unique(np.concatenate([truth[u], rng.integers(0, 2, size=(6 * H, M)).astype(np.int8)]), axis=0)
        rng.shuffle(pool)
        haps[u] = pool[:H]
    counts = rng.integers(1, 4, size=(U, R)).astype(np.int64)
    return reads, haps, counts, rng


@pytest.mark.parametrize("step_type", ["Gibbs", "Metropolis-Hastings"])
@pytest.mark.parametrize("K,H,M,R", [(4, 6, 6, 40), (2, 9, 5, 20), (6, 5, 4, 70), (3, 12, 6, 130), (4, 18, 6, 1400),  # 202 KB of products per chain
                                     (10, 6, 6, 50), (12, 5, 5, 90), (15, 4, 6, 40), (9, 7, 100, 60)],
                         # (round 5: ploidies 9 to 15, and known haplotypes of 100 SNVs)
                         ids=["K4", "K2", "K6", "K3", "tables-in-workspace", "K10", "K12", "K15", "K9-100snvs"])
def test_traces_match_oracle_step_for_step(step_type, K, H, M, R):
    from mchap_amd.calling_mcmc import CallingMCMC

    U = 3
    reads, haps, counts, rng = _inputs(U, K, H, M, R, seed=K * 100 + H)
    F = np.array([0.0, 0.12, 0.3])
    fr = rng.dirichlet(np.ones(H), size=U)
    st = 0 if step_type == "Gibbs" else 1
    for prior, rc in ((None, None), ((F, None), counts), ((F, fr), counts)):
        model = CallingMCMC(ploidy=K, haplotypes=haps[0], prior=None, steps=80, chains=2, random_seed=17, step_type=step_type)
        traces = model.fit_batch(reads, rc, haplotypes=haps, prior=prior)
        for u in range(U):
            pr = None if prior is None else (float(F[u]), None if prior[1] is None else fr[u])
            g, l = orc.call_mcmc(reads[u], haps[u], K, steps=80, chains=2, step_type=st, read_counts=None if rc is None else rc[u],
                                 prior=pr, rng_kind=orc.RNG_PHILOX, seed=17, stream_id=u)
            assert np.array_equal(traces[u].genotypes, g), (u, prior is None)
            np.testing.assert_allclose(traces[u].llks, l, rtol=1e-10, atol=1e-9)
            assert traces[u].n_allele == H


def test_initial_genotype_zero_reads_and_errors():
    from mchap_amd.calling_mcmc import CallingMCMC

    reads, haps, counts, rng = _inputs(2, 4, 5, 5, 30, seed=5)
    ini = np.sort(rng.integers(0, 5, size=(2, 4)), axis=1)
    model = CallingMCMC(ploidy=4, haplotypes=haps[0], steps=40, chains=3, random_seed=3)
    traces = model.fit_batch(reads, None, initial=ini, haplotypes=haps)
    for u in range(2):
        g, l = orc.call_mcmc(reads[u], haps[u], 4, steps=40, chains=3, initial=ini[u], rng_kind=orc.RNG_PHILOX, seed=3, stream_id=u)
        assert np.array_equal(traces[u].genotypes, g)
    # the one-unit operator form == unit 0 of the batch
    one = CallingMCMC(ploidy=4, haplotypes=haps[0], steps=40, chains=3, random_seed=3).fit(reads[0], initial=ini[0])
    assert np.array_equal(one.genotypes, traces[0].genotypes)
    # no reads: every genotype equally likely under the flat prior; llk 0
    empty = CallingMCMC(ploidy=4, haplotypes=haps[0], steps=30, chains=2, random_seed=1).fit(np.empty((0, 5, 2)))
    assert np.all(empty.llks == 0.0)
    with pytest.raises(ValueError):
        CallingMCMC(ploidy=4, haplotypes=haps[0], step_type="Other").fit(reads[0])
    # no variants: the constant trace (calling/classes.py:77-83)
    none = CallingMCMC(ploidy=4, haplotypes=np.zeros((1, 0), np.int8), steps=10, chains=2).fit(np.empty((5, 0, 0)))
    assert none.genotypes.shape == (2, 10, 4) and np.isnan(none.llks).all()


@pytest.mark.parametrize("chains", [1, 4, 5, 9])
def test_chains_of_a_unit_share_its_tables(chains):
    """The chains of a unit run as the wavefronts of one workgroup over shared LDS tables (up to four per workgroup: five chains are
    a full workgroup and one of a single chain whose other wavefronts leave at once): every chain's trace equals the oracle's."""
    from mchap_amd.calling_mcmc import CallingMCMC

    reads, haps, counts, rng = _inputs(3, 4, 7, 6, 90, seed=40 + chains)
    for step_type, st in (("Gibbs", 0), ("Metropolis-Hastings", 1)):
        traces = CallingMCMC(ploidy=4, haplotypes=haps[0], prior=None, steps=60, chains=chains, random_seed=9, step_type=step_type).fit_batch(
            reads, None, haplotypes=haps, prior=(np.full(3, 0.1), None))
        for u in range(3):
            g, l = orc.call_mcmc(reads[u], haps[u], 4, steps=60, chains=chains, step_type=st, prior=(0.1, None), rng_kind=orc.RNG_PHILOX, seed=9, stream_id=u)
            assert np.array_equal(traces[u].genotypes, g)
            np.testing.assert_allclose(traces[u].llks, l, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("prior", [None, (0.2, None)])
def test_gibbs_and_mh_agree_with_the_exact_posterior(prior):
    """25 000 steps x 2 chains of either step type against genotype_posteriors over all genotypes (2 decimals)."""
    from mchap_amd import calling
    from mchap_amd.calling_mcmc import CallingMCMC

    K, H = 4, 4
    reads, haps, counts, rng = _inputs(1, K, H, 4, 14, seed=31, qual=(3, 10))
    llks = calling.genotype_likelihoods(reads[0], K, haps[0]).astype(np.float64)
    exact = calling.genotype_posteriors(llks, K, H, prior)
    for step_type in ("Gibbs", "Metropolis-Hastings"):
        model = CallingMCMC(ploidy=K, haplotypes=haps[0], prior=prior, steps=25000, chains=2, random_seed=11, step_type=step_type)
        post = model.fit(reads[0]).burn(1000).posterior().as_array(H)
        assert np.abs(post - exact).max() < 0.015, step_type


@pytest.mark.parametrize("lanes", ["default", "0", "1", "5", "13"])
@pytest.mark.parametrize("shape", ["settled", "wandering", "octoploid", "many-haplotypes"])
def test_settled_chains_a_lane_each_give_the_same_traces(shape, lanes, monkeypatch):
    """Round 5: a chain whose step met only remembered contexts goes to call_coast_kernel (a lane per chain) and comes back at the
    first context it does not know -- settled chains (no coming back), chains of few low-quality reads that keep moving (every
    round used up, the last launch finishes them), ploidy 8 (seven alleles in a context's key), 40 known haplotypes; whatever
    the number of lanes in use per wavefront (MCHAP_HIP_CALL_LANES; 0: no hand-over), the oracle's traces step for step."""
    from mchap_amd.calling_mcmc import CallingMCMC

    K, H, M, R, qual, steps = {"settled": (4, 16, 8, 200, (5, 25), 400), "wandering": (4, 8, 6, 12, (2, 8), 700),
                               "octoploid": (8, 6, 5, 60, (5, 25), 300), "many-haplotypes": (3, 40, 8, 30, (3, 12), 300)}[shape]
    if lanes != "default":
        monkeypatch.setenv("MCHAP_HIP_CALL_LANES", lanes)
    U = 5
    reads, haps, counts, rng = _inputs(U, K, H, M, R, seed=7 * K + H, qual=qual)
    traces = CallingMCMC(ploidy=K, haplotypes=haps[0], prior=None, steps=steps, chains=3, random_seed=23).fit_batch(
        reads, None, haplotypes=haps, prior=(np.full(U, 0.1), None))
    moves = 0
    for u in range(U):
        g, l = orc.call_mcmc(reads[u], haps[u], K, steps=steps, chains=3, step_type=0, prior=(0.1, None), rng_kind=orc.RNG_PHILOX, seed=23, stream_id=u)
        assert np.array_equal(traces[u].genotypes, g), u
        np.testing.assert_allclose(traces[u].llks, l, rtol=1e-10, atol=1e-9)
        moves += int((g[:, 1:] != g[:, :-1]).any(axis=-1).sum())
    if shape == "wandering":
        assert moves > U * 3 * steps // 10, moves  # (the shape does what its name says)


@pytest.mark.parametrize("shape", ["settled", "wandering", "octoploid", "one-chain", "ploidy-10"])
def test_device_summaries_of_call_traces_equal_the_host_classes(shape):
    """Round 5: CallingMCMC.fit_batch_summaries leaves the traces in HBM and summarises them there (trace_posterior_kernel; the
    calling classes' replicate incongruence by mchap_call_incongruence_*_device) -- against GenotypeAllelesMultiTrace.burn()
    .posterior(), .mode(genotype_support=True), .replicate_incongruence(), .posterior_frequencies() of fit_batch's traces of the
    same seed (calling/classes.py:166-284, 303-362): genotype order, probabilities, mode genotype, its probability, the
    support's, MCI, allele frequencies / counts / occurrence, and the GP array.  Chains of few poor reads visit more genotypes
    than the batch launch keeps (listed launch); ploidy 10 takes the host classes (the device summary's ploidy bound is 8)."""
    from mchap_amd.calling_mcmc import CallingMCMC

    K, H, M, R, qual, steps, chains = {"settled": (4, 12, 8, 150, (5, 25), 300, 2), "wandering": (4, 30, 6, 3, (2, 6), 900, 3),
                                       "octoploid": (8, 6, 5, 20, (3, 12), 300, 2), "one-chain": (3, 7, 5, 40, (5, 25), 200, 1),
                                       "ploidy-10": (10, 4, 4, 30, (5, 25), 120, 2)}[shape]
    U, burn = 4, steps // 3
    reads, haps, counts, rng = _inputs(U, K, H, M, R, seed=11 * K + H, qual=qual)
    model = CallingMCMC(ploidy=K, haplotypes=haps[0], prior=None, steps=steps, chains=chains, random_seed=29)
    kw = dict(haplotypes=haps, prior=(np.full(U, 0.15), None))
    got = model.fit_batch_summaries(reads, None, burn=burn, incongruence_threshold=0.6, **kw)
    traces = model.fit_batch(reads, None, **kw)
    most = 0
    for u in range(U):
        tr = traces[u].burn(burn)
        post = tr.posterior()
        assert np.array_equal(got[u].genotypes, post.genotypes), u
        np.testing.assert_array_equal(got[u].counts / got[u].n_obs, post.probabilities)
        alleles, gprob, sprob = post.mode(genotype_support=True)
        assert np.array_equal(got[u].alleles, alleles) and abs(got[u].gprob - gprob) < 1e-15 and abs(got[u].sprob - sprob) < 1e-12
        assert got[u].mci == tr.replicate_incongruence(threshold=0.6)
        for a, b in zip(got[u].posterior_frequencies(), tr.posterior_frequencies()):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(got[u].posterior().as_array(H), post.as_array(H))
        labels = np.arange(H) * 2 + 1   # (--filter-input-haplotypes: alleles renamed by an increasing map)
        np.testing.assert_array_equal(got[u].relabel(labels).posterior_frequencies()[2], tr.relabel(labels).posterior_frequencies()[2])
        most = max(most, len(post.probabilities))
    if shape == "wandering":
        assert most > 512, most
