"""GPU parity at the FULL sizes of BASELINE.json's configs that the other GPU tests only touch at reduced shapes.

configs[3] (call-exact): hexaploid, 16 known haplotypes over 10 SNVs, 500 reads -> G = C(21, 6) = 54 264 genotypes per
unit (14 enumeration blocks per unit, multi-block arg-max / log-sum-exp merge), Dirichlet-multinomial prior with
inbreeding 0.1 and Dirichlet(1) allele frequencies.  Reference: calling/exact.py:156-329.
configs[4] (LDS-pressure sampler): octoploid, 20 SNVs, 1000 reads, 4 chains; and its secondary variant of SURVEY.md 8d,
one chain under the temperature ladder (0.001, 0.01, 0.1, 1.0).  Reference: assemble/mcmc.py:268-426.

Checker: the CPU oracle (oracle/mchap_oracle.c) on the same inputs / Philox streams.  Tolerances as in DESIGN.md:
integers bit-exact, fp64 1e-9 relative (llk traces 1e-10), float32 likelihoods 2.5e-7 relative, posteriors formed with
float32 arithmetic 3e-5 relative."""
import numpy as np
import pytest

from oracle import binding as orc

pytestmark = pytest.mark.gpu


def _config4_inputs(U, seed=4):
    from mchap_amd.synth import synth_units

    K, H, M, R = 6, 16, 10, 500
    rng = np.random.default_rng(seed)
    reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(5, 10), first_unit=400)
    # odd units: very low base qualities, so that the posterior is spread over many genotypes (the even units' 500
    # good reads concentrate it on one)
    noisy, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(5, 10), first_unit=400, qual=(1, 4))
    reads[1::2] = noisy[1::2]
    haps = np.zeros((U, H, M), np.int8)
    for u in range(U):
        # the unit's true haplotypes plus random ones, 16 distinct in a random order
        pool = np.unique(np.concatenate([truth[u], rng.integers(0, 2, size=(8 * H, M)).astype(np.int8)]), axis=0)
        rng.shuffle(pool)
        keep = [p for p in pool if not any(np.array_equal(p, t) for t in truth[u])][: H - len(np.unique(truth[u], axis=0))]
        hs = np.concatenate([np.unique(truth[u], axis=0), np.array(keep, np.int8)])
        rng.shuffle(hs)
        assert len(hs) == H
        haps[u] = hs
    F = np.full(U, 0.1)
    freqs = rng.dirichlet(np.ones(H), size=U)
    return K, H, M, R, reads, haps, F, freqs


def test_config4_posterior_mode_full_size():
    """posterior_mode_batch at configs[3]: every returned statistic against the oracle, two units."""
    from mchap_amd import calling

    U = 2
    K, H, M, R, reads, haps, F, freqs = _config4_inputs(U)
    assert calling.count_unique_genotypes(H, K) == 54264
    out = calling.posterior_mode_batch(reads, K, haps, None, (F, freqs), True, True, True)
    for u in range(U):
        a, ml, mp, sp, fq, oc = orc.posterior_mode(reads[u], K, haps[u], None, (float(F[u]), freqs[u]))
        assert out[0][u].tolist() == a.tolist()
        np.testing.assert_allclose([out[1][u], out[2][u], out[3][u]], [ml, mp, sp], rtol=1e-9)
        np.testing.assert_allclose(out[4][u], fq, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(out[5][u], oc, rtol=1e-9, atol=1e-300)
        assert abs(out[4][u].sum() - 1.0) < 1e-9
    # the single-unit operator form gives the same answer as the batch
    one = calling.posterior_mode(reads[1], K, haps[1], None, (float(F[1]), freqs[1]), True, True, True)
    assert one[0].tolist() == out[0][1].tolist()
    np.testing.assert_allclose(one[1:4], [out[1][1], out[2][1], out[3][1]], rtol=1e-12)


def test_config4_likelihood_and_posterior_arrays_full_size():
    """genotype_likelihoods (float32 store) and genotype_posteriors (float32 arithmetic) at configs[3], plus the array
    summaries the reference's --report GL/GP path derives from them (call_exact.py:126-159)."""
    from mchap_amd import calling

    U = 2
    K, H, M, R, reads, haps, F, freqs = _config4_inputs(U)
    for u in range(U):
        prior = (float(F[u]), freqs[u])
        l32 = calling.genotype_likelihoods(reads[u], K, haps[u])
        assert l32.dtype == np.float32 and l32.shape == (54264,)
        e32, e64 = orc.genotype_likelihoods(reads[u], K, haps[u], None)
        np.testing.assert_allclose(l32, e32, rtol=2.5e-7)
        post = calling.genotype_posteriors(e32, K, H, prior)
        ref = orc.genotype_posteriors(e32, K, H, prior)
        # float32 arithmetic (calling/exact.py:317, jitutils.py:7-74): the normaliser is a 54 264-term log-sum-exp in
        # float32, whose value depends on the summation order by a few units in the last place of a number of
        # magnitude |llk| (4 500 for the noisy unit: one ulp = 4.9e-4 in the log domain, i.e. 4.9e-4 relative in
        # every probability).  Tolerance: 4 ulp of float32 at that magnitude, never below the 3e-5 of the small cases.
        ulp = float(np.spacing(np.float32(np.abs(e32).max())))
        np.testing.assert_allclose(post, ref, rtol=max(3e-5, 4 * ulp), atol=1e-12)
        post64 = calling.genotype_posteriors(e64, K, H, prior)
        ref64 = orc.genotype_posteriors(e64, K, H, prior)
        np.testing.assert_allclose(post64, ref64, rtol=1e-9, atol=1e-300)
        assert abs(post64.sum() - 1.0) < 1e-9
        f, c, o = calling.posterior_allele_frequencies(ref64, K, H)
        rf, rc, ro = orc.posterior_allele_frequencies(ref64, K, H)
        np.testing.assert_allclose(np.stack([f, c, o]), np.stack([rf, rc, ro]), rtol=1e-9, atol=1e-300)
        # mode of the array == streaming mode (same genotype; GP of it within the float64 tolerance)
        idx = int(np.argmax(post64))
        mode = calling.posterior_mode(reads[u], K, haps[u], None, prior, True)
        assert calling.index_as_genotype_alleles(idx, K).tolist() == mode[0].tolist()
        np.testing.assert_allclose(post64[idx], mode[2], rtol=1e-6)
        ag, ap = calling.alternate_dosage_posteriors(mode[0], post64)
        np.testing.assert_allclose(ap.sum(), mode[3], rtol=1e-6)


def _config5_oracle(reads, steps, chains, temperatures, seed):
    from mchap_amd.classes import sort_haplotypes
    from tests.helpers import beta_break_table

    ref = []
    for u, rd in enumerate(reads):
        # (the oracle's llk cache is the reference's own: same values, a quarter of the time at this shape)
        cfg = orc.make_cfg(8, steps, chains, None, temperatures, llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX,
                           seed=seed, stream_id=u, break_table=beta_break_table(20, 1.0, 3.0))
        g, l, code = orc.denovo_fit(cfg, rd, [2] * 20)
        assert code == 0
        ref.append((sort_haplotypes(g), l))
    return ref


def test_config5_full_shape_four_chains_300_steps(monkeypatch):
    """configs[4]: K = 8, 20 SNVs, 1000 reads, 4 parallel chains -- 300 steps of every chain step for step against the
    oracle on the kernels the benchmark number rests on: the default dispatch (0), the phased sampler named explicitly (5)
    and the speculative kernel (3).  The window holds the hand-over at step 16, four coasting sweeps of 64 steps and, for
    the second unit (200 reads of base quality 5..15: its chains move 7-25 times in these 300 steps, the last time after
    step 230), hand-backs, resume rounds and a second table completion."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    clean, _, _ = synth_units(1, ploidy=8, n_pos=20, n_reads=1000, window=(8, 20), first_unit=77)
    noisy, _, _ = synth_units(1, ploidy=8, n_pos=20, n_reads=200, window=(8, 20), first_unit=78, qual=(5, 15))
    reads = [clean[0], noisy[0]]
    kw = dict(chains=4, temperatures=(1.0,))
    ref = _config5_oracle(reads, 300, 4, (1.0,), 11)
    moves = [int((np.diff(ref[u][1], axis=1) != 0).sum()) for u in range(2)]
    late = int(max(np.flatnonzero(np.diff(ref[1][1], axis=1).any(axis=0))))
    assert moves[1] >= 12 and late > 150, (moves, late)  # the noisy unit does exercise the hand-back path
    # (tuning flag 512: without the product rows of deep units in the workspace -- the plain instantiation, every factor of
    # every read formed for every request, as before round 3)
    for kernel, flags in ((0, 0), (5, 0), (3, 0), (0, 512)):
        monkeypatch.setenv("MCHAP_HIP_KERNEL", str(kernel))
        monkeypatch.setenv("MCHAP_HIP_FLAGS", str(flags))
        model = DenovoMCMC(ploidy=8, n_alleles=[2] * 20, steps=300, random_seed=11, **kw)
        traces = model.fit_batch(reads)
        assert ("phased" in model.last_sampler) == (kernel != 3), model.last_sampler
        for u, tr in enumerate(traces):
            assert tr.genotypes.shape == (4, 300, 8, 20)
            assert np.array_equal(tr.genotypes, ref[u][0]), "kernel %d unit %d" % (kernel, u)
            np.testing.assert_allclose(tr.llks, ref[u][1], rtol=1e-10)


def test_config5_full_shape_tempered(monkeypatch):
    """configs[4]'s secondary variant: one chain under the temperature ladder (0.001, 0.01, 0.1, 1.0), step for
    step against the oracle: 200 steps through the default dispatch (round 4: the replicas side by side, one wavefront each) --
    the hot replicas move at every other sub-step, so this is some 60 000 accepted moves and 800 swap attempts per unit --, and
    the first 25 of them on the one-wavefront form of the speculative kernel and on the lanes-over-chains kernel (draws are
    numbered per step: a shorter run is a prefix of a longer one)."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(2, ploidy=8, n_pos=20, n_reads=1000, window=(8, 20), first_unit=77)
    temps = (0.001, 0.01, 0.1, 1.0)
    ref = _config5_oracle(list(reads), 200, 1, temps, 11)
    for kernel, flags, steps in ((0, "0", 200), (3, "16384", 25), (2, "0", 25)):
        monkeypatch.setenv("MCHAP_HIP_KERNEL", str(kernel))
        monkeypatch.setenv("MCHAP_HIP_FLAGS", flags)
        model = DenovoMCMC(ploidy=8, n_alleles=[2] * 20, steps=steps, random_seed=11, chains=1, temperatures=temps)
        traces = model.fit_batch(list(reads))
        for u, tr in enumerate(traces):
            assert tr.genotypes.shape == (1, steps, 8, 20)
            assert np.array_equal(tr.genotypes, ref[u][0][:, :steps]), "kernel %d unit %d" % (kernel, u)
            np.testing.assert_allclose(tr.llks, ref[u][1][:, :steps], rtol=1e-10)


def test_config2_default_dispatch_1000_steps_against_the_oracle(monkeypatch):
    """BASELINE.json configs[1] as benchmarked: tetraploid, 8 SNVs, 200 reads, 1000 steps x 2 chains through the library's
    DEFAULT dispatch (the phased sampler: first phase, table completion, coasting, resume rounds) -- every step of every
    chain of 64 loci against the oracle on the same Philox streams; the device-side posterior summary (mode genotype,
    GPM, SPM) against the reference's classes on the oracle's trace."""
    import torch

    from mchap_amd import DenovoMCMC, GenotypeMultiTrace
    from mchap_amd.assemble import break_table, unpack_trace
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    monkeypatch.delenv("MCHAP_HIP_KERNEL", raising=False)
    U = 64
    reads, _, _ = synth_units(U)
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42)
    batch = DenovoDeviceBatch(model, reads)
    assert "phased" in batch.sampler_name and "coast" in batch.sampler_name
    batch.run()
    batch.posterior(500)
    torch.cuda.synchronize()
    words, fixed, llks, status = batch.traces()
    assert (status == 0).all()
    cfg = orc.make_cfg(4, 1000, 2, None, (1.0,), llk_cache_threshold=100, seed=42, rng_kind=orc.RNG_PHILOX,
                       break_table=break_table(8, 1.0, 3.0))
    g, l, code, _ = orc.denovo_fit_batch(cfg, reads, [2] * 8, n_threads=0, keep_traces=True)
    assert code == 0
    post = batch.posterior_host()
    for u in range(U):
        ref = sort_haplotypes(g[u])
        assert np.array_equal(batch.genotypes(u, words, fixed), ref), "unit %d" % u
        np.testing.assert_allclose(llks[u], l[u], rtol=1e-10)
        sup = GenotypeMultiTrace._from_sorted(ref, l[u]).burn(500).posterior().mode_genotype_support()
        mg, gpm = sup.mode_genotype()
        assert np.array_equal(unpack_trace(post["mode_words"][u][None], fixed[u], 2)[0], mg)
        assert abs(post["stats"][u][1] - gpm) < 1e-12 and abs(post["stats"][u][0] - sup.probabilities.sum()) < 1e-12


def test_config4_call_exact_arrays_batched_on_the_device():
    """`mchap call-exact --report GL GP` at configs[3] for a chunk of 24 (locus x sample) units in one device call: no
    host loop over units or genotypes (the reference's per-sample sequence genotype_likelihoods -> genotype_posteriors ->
    argmax -> alternate_dosage_posteriors -> posterior_allele_frequencies, call_exact.py:126-159).  Two units against
    the oracle, every unit against the streaming form and the invariants of a distribution."""
    from mchap_amd import calling
    from mchap_amd.device import ExactDeviceBatch

    U = 24
    K, H, M, R, reads, haps, F, freqs = _config4_inputs(U)
    batch = ExactDeviceBatch(reads, K, haps, None, (F, freqs))
    batch.run(streaming=True, arrays=True)
    mode = batch.mode_results()
    arr = batch.array_results()
    assert arr["posteriors"].shape == (U, 54264) and arr["llks"].dtype == np.float32
    np.testing.assert_allclose(arr["posteriors"].sum(axis=1), 1.0, rtol=2e-3)   # float32 arithmetic at |llk| ~ 4500
    np.testing.assert_allclose(arr["freqs"].sum(axis=1), arr["posteriors"].sum(axis=1), rtol=1e-9)
    np.testing.assert_allclose(mode[4].sum(axis=1), 1.0, rtol=1e-9)
    for u in range(U):
        # array mode == streaming mode whenever the mode is not a near tie (float32 vs float64 arithmetic)
        if mode[2][u] > 0.6:
            assert arr["alleles"][u].tolist() == mode[0][u].tolist()
            np.testing.assert_allclose(arr["prob"][u], mode[2][u], rtol=2e-3)
            np.testing.assert_allclose(arr["support_prob"][u], mode[3][u], rtol=2e-3)
    for u in (0, 1):
        prior = (float(F[u]), freqs[u])
        e32, _ = orc.genotype_likelihoods(reads[u], K, haps[u], None)
        np.testing.assert_allclose(arr["llks"][u], e32, rtol=2.5e-7)
        ref = orc.genotype_posteriors(e32, K, H, prior)
        ulp = float(np.spacing(np.float32(np.abs(e32).max())))
        np.testing.assert_allclose(arr["posteriors"][u], ref, rtol=max(3e-5, 4 * ulp), atol=1e-12)
        a, ml, mp, sp, fq, oc = orc.posterior_mode(reads[u], K, haps[u], None, prior)
        assert mode[0][u].tolist() == a.tolist()
        np.testing.assert_allclose([mode[1][u], mode[2][u], mode[3][u]], [ml, mp, sp], rtol=1e-9)
        np.testing.assert_allclose(mode[4][u], fq, rtol=1e-9, atol=1e-300)


WIDE_SHAPES = [
    # (ploidy, SNVs, reads, steps, first synthetic unit): ploidy x SNVs in 129..192 -- three sub-steps per lane of the
    # one-chain-per-wavefront kernels (two cover 128; before round 3 only the octoploid instantiation had a third)
    (4, 40, 120, 120, 500),
    (4, 44, 60, 100, 510),
    (6, 30, 150, 100, 520),
    (5, 38, 100, 100, 530),
    (3, 44, 90, 120, 540),
]


@pytest.mark.parametrize("shape", WIDE_SHAPES, ids=lambda s: "K%d-M%d" % (s[0], s[1]))
def test_wide_loci_on_the_fast_samplers(shape, monkeypatch):
    """Loci with many SNVs (33-44 at ploidy 3-6: 129-192 mutation sub-steps per step) run on the phased sampler and on the
    speculative kernel -- not on the general lanes-over-chains kernel they fell to before -- and every step of every chain
    equals the oracle's: a clean unit (settles: hand-over, table completion over 780-990 intervals, coasting) and a shallow
    low-quality one (keeps moving: speculation windows over three slots per lane).  Reference: assemble/mcmc.py:268-426,
    mutation.py:164-246 (no limit on the number of positions)."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units
    from tests.helpers import beta_break_table

    K, M, R, steps, first = shape
    clean, _, _ = synth_units(1, ploidy=K, n_pos=M, n_reads=R, window=(M // 3, M), first_unit=first)
    noisy, _, _ = synth_units(1, ploidy=K, n_pos=M, n_reads=24, window=(M // 3, M), first_unit=first + 1, qual=(4, 14))
    reads = [clean[0], noisy[0]]
    ref = []
    for u, rd in enumerate(reads):
        cfg = orc.make_cfg(K, steps, 2, None, (1.0,), llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX, seed=5, stream_id=u,
                           break_table=beta_break_table(M, 1.0, 3.0))
        g, l, code = orc.denovo_fit(cfg, rd, [2] * M)
        assert code == 0
        ref.append((sort_haplotypes(g), l))
    assert (np.diff(ref[1][1], axis=1) != 0).sum() >= 10  # the noisy unit does move
    for kernel, name in ((0, "phased"), (3, "denovo_spec_kernel<%d, 64>" % K), (2, "simt")):
        monkeypatch.setenv("MCHAP_HIP_KERNEL", str(kernel))
        model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=steps, chains=2, random_seed=5)
        traces = model.fit_batch(reads)
        assert name in model.last_sampler, model.last_sampler
        for u, tr in enumerate(traces):
            assert tr.genotypes.shape == (2, steps, K, M)
            assert np.array_equal(tr.genotypes, ref[u][0]), "kernel %d unit %d" % (kernel, u)
            np.testing.assert_allclose(tr.llks, ref[u][1], rtol=1e-10)
