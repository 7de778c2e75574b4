"""GPU parity: the HIP sampler against the CPU oracle on the same Philox streams.

Integer traces must be bit-exact, log-likelihoods within 1e-10 relative (fp64; the kernel sums reads in a
different order).  All calls go through the C ABI (libmchap_hip.so)."""
import numpy as np
import pytest

from oracle import binding as orc
from tests.helpers import beta_break_table

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[5, 4, 3, 2, 1], ids=["phased", "steady-state", "speculative", "lanes-over-chains", "wave-per-chain"])
def sampler_kernel(request, monkeypatch):
    """Every test runs against both sampler kernels; they must give identical traces."""
    monkeypatch.setenv("MCHAP_HIP_KERNEL", str(request.param))
    if request.param in (1, 4):  # the two earlier designs live in the parity suite's library only
        monkeypatch.setenv("MCHAP_HIP_TEST_KERNELS", "1")
    return request.param


def _oracle_trace(model, reads, n_alleles, rc, stream_id, initial=None, inbreeding="model", ploidy=None):
    F = model.inbreeding if inbreeding == "model" else inbreeding
    M = reads.shape[1]
    cfg = orc.make_cfg(model.ploidy if ploidy is None else ploidy, model.steps, model.chains, F, model.temperatures,
                       fix_homozygous=model.fix_homozygous, p_recomb=model.recombination_step_probability,
                       p_partial_dosage=model.partial_dosage_step_probability, p_dosage=model.dosage_step_probability,
                       n_intervals=model.n_intervals, llk_cache_threshold=-1, rng_kind=orc.RNG_PHILOX,
                       seed=model.random_seed, stream_id=stream_id, break_table=beta_break_table(M, model.alpha, model.beta))
    g, l, code = orc.denovo_fit(cfg, reads, n_alleles, rc, initial)
    assert code == 0
    return g, l


def _check(model, reads_list, counts_list=None, initial=None, **kw):
    from mchap_amd.classes import sort_haplotypes

    traces = model.fit_batch(reads_list, counts_list, initial, **kw)
    for u, tr in enumerate(traces):
        rc = None if counts_list is None else counts_list[u]
        ini = None if initial is None else initial[u]
        F = "model" if kw.get("inbreeding") is None else kw["inbreeding"][u]
        K = None if kw.get("ploidy") is None else kw["ploidy"][u]
        g, l = _oracle_trace(model, reads_list[u], model.n_alleles, rc, u, ini, F, K)
        assert tr.genotypes.dtype == np.int8
        assert np.array_equal(tr.genotypes, sort_haplotypes(g)), "unit %d" % u
        np.testing.assert_allclose(tr.llks, l, rtol=1e-10, atol=1e-9, equal_nan=True)


def test_config2_shape_small_batch():
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(6, ploidy=4, n_pos=8, n_reads=200)
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=150, chains=2, random_seed=42)
    _check(model, list(reads))


@pytest.mark.parametrize("F", [None, 0.0, 0.1])
@pytest.mark.parametrize("temps", [(1.0,), (0.2, 0.6, 1.0)])
def test_priors_and_tempering(F, temps):
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(3, ploidy=4, n_pos=6, n_reads=48, first_unit=100)
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 6, inbreeding=F, steps=120, chains=2, temperatures=temps, random_seed=7)
    _check(model, list(reads))


@pytest.mark.parametrize("K,M,A,R", [(2, 5, 2, 30), (6, 4, 2, 70), (8, 5, 2, 100), (4, 5, 3, 40), (3, 3, 4, 20), (4, 12, 2, 300)])
def test_shapes(K, M, A, R):
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(2, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=7, window=(2, M))
    model = DenovoMCMC(ploidy=K, n_alleles=[A] * M, inbreeding=0.05, steps=60, chains=3, random_seed=11)
    _check(model, list(reads))


def test_read_counts_dedup_and_ragged_batch():
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import dedup_unit, synth_units

    reads, _, _ = synth_units(4, ploidy=4, n_pos=8, n_reads=200, dedup=True, first_unit=50)
    rl, cl = zip(*[dedup_unit(r) for r in reads])
    assert len({len(r) for r in rl}) > 1  # ragged
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=100, chains=2, random_seed=3)
    _check(model, list(rl), list(cl))


def test_fixed_homozygous_zero_reads_initial_and_options():
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import as_probabilistic, synth_units

    reads, calls, _ = synth_units(3, ploidy=4, n_pos=6, n_reads=150, first_unit=9)
    # unit 0: two positions made homozygous; unit 1: as is; unit 2: zero reads
    c0 = calls[0].copy()
    c0[:, 1] = np.where(c0[:, 1] >= 0, 0, -1)
    c0[:, 4] = np.where(c0[:, 4] >= 0, 1, -1)
    r0 = as_probabilistic(c0, np.full(6, 2), 0.99)
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 6, steps=80, chains=2, random_seed=5, n_intervals=2,
                       recombination_step_probability=0.3, partial_dosage_step_probability=0.8, dosage_step_probability=0.6)
    rl = [r0, reads[1], np.empty((0, 6, 2))]
    traces = model.fit_batch(rl)
    assert (traces[0].genotypes[..., 1] == 0).all() and (traces[0].genotypes[..., 4] == 1).all()
    _check(model, rl)
    # all positions fixed -> constant trace, NaN llks (reference assemble/mcmc.py:189-199)
    c_all = np.zeros((100, 3), dtype=np.int8)
    c_all[:, 1] = 1
    tr = DenovoMCMC(ploidy=4, n_alleles=[2, 2, 2], steps=10, chains=2, random_seed=1).fit(as_probabilistic(c_all, np.full(3, 2), 0.99))
    assert np.isnan(tr.llks).all()
    assert (tr.genotypes == np.array([0, 1, 0], dtype=np.int8)).all()
    # user-supplied initial state
    rng = np.random.default_rng(0)
    ini = rng.integers(0, 2, size=(2, 4, 6)).astype(np.int8)
    model2 = DenovoMCMC(ploidy=4, n_alleles=[2] * 6, steps=50, chains=2, random_seed=9, fix_homozygous=2.0)
    _check(model2, [reads[1]], None, [ini])


def test_per_unit_ploidy_and_inbreeding_and_stream_ids():
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    r4, _, _ = synth_units(2, ploidy=4, n_pos=6, n_reads=60, first_unit=30)
    r2, _, _ = synth_units(1, ploidy=2, n_pos=6, n_reads=60, first_unit=40)
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 6, steps=60, chains=2, random_seed=21)
    rl = [r4[0], r2[0], r4[1]]
    _check(model, rl, ploidy=[4, 2, 4], inbreeding=[None, 0.3, 0.0])
    # results depend on (seed, stream id) only, not on batch composition
    a = model.fit_batch(rl, ploidy=[4, 2, 4], inbreeding=[None, 0.3, 0.0])
    b = model.fit_batch([rl[2]], ploidy=[4], inbreeding=[0.0], stream_ids=[2])
    assert np.array_equal(a[2].genotypes, b[0].genotypes) and np.array_equal(a[2].llks, b[0].llks)


def test_errors():
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(1, ploidy=4, n_pos=4, n_reads=20)
    with pytest.raises(AssertionError):
        DenovoMCMC(ploidy=4, n_alleles=[2] * 4, temperatures=(0.5, 0.9), random_seed=1).fit(reads[0])
    with pytest.raises(ValueError):
        DenovoMCMC(ploidy=4, n_alleles=[2] * 4, n_intervals=6, steps=5, random_seed=1).fit(reads[0])
    with pytest.raises(NotImplementedError):  # (ploidies 9 to 15 run since round 4: tests/test_gpu_wide.py)
        DenovoMCMC(ploidy=16, n_alleles=[2] * 4, steps=5, random_seed=1).fit(reads[0])
    import os

    if os.environ.get("MCHAP_HIP_KERNEL") != "1":  # (kernel 1 of the parity suite stays at ploidy 8)
        assert DenovoMCMC(ploidy=12, n_alleles=[2] * 4, steps=5, random_seed=1).fit(reads[0]).genotypes.shape == (2, 5, 12, 4)


def test_log_likelihood_hook():
    import ctypes as C
    from mchap_amd import _lib
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(1, ploidy=4, n_pos=8, n_reads=200)
    rng = np.random.default_rng(1)
    g = rng.integers(0, 2, size=(50, 4, 8)).astype(np.int8)
    out = np.zeros(50)
    rd = np.ascontiguousarray(reads[0])
    rc = rng.integers(1, 5, size=200).astype(np.int64)
    _lib.check(_lib.lib().mchap_log_likelihood_batch(_lib.ptr(rd), 200, 8, 2, _lib.ptr(rc), _lib.ptr(g), 50, 4, _lib.ptr(out)))
    expect = np.array([orc.log_likelihood(rd, gi, rc) for gi in g])
    np.testing.assert_allclose(out, expect, rtol=1e-12)


def test_llk_cache_is_results_neutral():
    """Reference tests/test_application_assemble.py:356,390: identical output with the llk cache on and off."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(4, ploidy=4, n_pos=8, n_reads=200, first_unit=77)
    for F, temps in ((None, (1.0,)), (0.1, (0.3, 1.0))):
        kw = dict(ploidy=4, n_alleles=[2] * 8, inbreeding=F, steps=200, chains=2, temperatures=temps, random_seed=13)
        on = DenovoMCMC(llk_cache_threshold=100, **kw).fit_batch(list(reads))
        off = DenovoMCMC(llk_cache_threshold=-1, **kw).fit_batch(list(reads))
        for a, b in zip(on, off):
            assert np.array_equal(a.genotypes, b.genotypes)
            assert np.array_equal(a.llks, b.llks)
    # wide genotypes take the hashed-key path (K * Mh * bits > 63)
    reads, _, _ = synth_units(2, ploidy=6, n_pos=14, n_reads=120, first_unit=5, window=(5, 14))
    kw = dict(ploidy=6, n_alleles=[2] * 14, steps=60, chains=2, random_seed=2)
    on = DenovoMCMC(llk_cache_threshold=100, **kw).fit_batch(list(reads))
    off = DenovoMCMC(llk_cache_threshold=-1, **kw).fit_batch(list(reads))
    for a, b in zip(on, off):
        assert np.array_equal(a.genotypes, b.genotypes) and np.array_equal(a.llks, b.llks)


def test_both_kernels_give_identical_traces(monkeypatch):
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    monkeypatch.setenv("MCHAP_HIP_TEST_KERNELS", "1")  # kernel 1 lives in the parity suite's library
    reads, _, _ = synth_units(70, ploidy=4, n_pos=8, n_reads=200, first_unit=500)  # > one wavefront of chains
    kw = dict(ploidy=4, n_alleles=[2] * 8, steps=120, chains=2, random_seed=99)
    a = DenovoMCMC(kernel=1, **kw).fit_batch(list(reads))
    for k in (2, 3):
        b = DenovoMCMC(kernel=k, **kw).fit_batch(list(reads))
        for x, y in zip(a, b):
            assert np.array_equal(x.genotypes, y.genotypes)
            np.testing.assert_allclose(x.llks, y.llks, rtol=1e-12)


def test_config5_shape_octoploid_deep_reads(sampler_kernel):
    """BASELINE.json configs[4] shape: K=8, 20 SNVs, 1000 reads.  K*M = 160 sub-steps: three per lane in the
    speculative kernel; the transposed table (320 KB) exceeds the prepare pass's LDS copy, and the LDS-staged kernel 1
    cannot hold it at all (it must say so)."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    if sampler_kernel == 1:
        reads, _, _ = synth_units(1, ploidy=8, n_pos=20, n_reads=1000, window=(8, 20))
        with pytest.raises(NotImplementedError):
            DenovoMCMC(ploidy=8, n_alleles=[2] * 20, steps=5, chains=2, random_seed=5).fit_batch(list(reads))
        return

    reads, _, _ = synth_units(2, ploidy=8, n_pos=20, n_reads=1000, window=(8, 20))
    model = DenovoMCMC(ploidy=8, n_alleles=[2] * 20, steps=25, chains=2, random_seed=5)
    _check(model, list(reads))


def test_continuous_probabilities_fall_back_to_float64_rows():
    """More than 256 distinct values in a unit's table: no dictionary, the evaluation reads float64 rows; a unit with
    few distinct values in the same batch uses its coded table.  Both must match the oracle."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    rng = np.random.default_rng(3)
    reads, _, _ = synth_units(3, ploidy=4, n_pos=6, n_reads=90, window=(3, 6))
    noisy = reads.copy()
    jitter = rng.uniform(0.9, 1.0, size=noisy.shape)
    noisy[1] = np.where(np.isnan(noisy[1]), np.nan, noisy[1] * jitter[1])  # unit 1: thousands of distinct values
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 6, steps=120, chains=2, random_seed=9)
    _check(model, list(noisy))


@pytest.mark.parametrize("n_pos", [17, 24])
def test_octoploid_three_substeps_per_lane(n_pos):
    """K = 8 with more than 128 sub-steps per mutation step (up to 192): third slot of the speculative kernel."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(2, ploidy=8, n_pos=n_pos, n_reads=90, window=(6, n_pos))
    model = DenovoMCMC(ploidy=8, n_alleles=[2] * n_pos, steps=40, chains=2, random_seed=n_pos)
    _check(model, list(reads))


@pytest.mark.parametrize("knobs", [{}, {"MCHAP_HIP_PIPE_FIRST": "1", "MCHAP_HIP_PIPE_RESUME": "1"}, {"MCHAP_HIP_ROUNDS": "0"},
                                   {"MCHAP_HIP_PIPE_FIRST": "2", "MCHAP_HIP_ROUNDS": "6", "MCHAP_HIP_PIPE_MAX": "2"},
                                   {"MCHAP_HIP_PIPE_GROUP": "16"}, {"MCHAP_HIP_PIPE_GROUP": "32", "MCHAP_HIP_PIPE_FIRST": "3"},
                                   {"MCHAP_HIP_PIPE_PARTS": "1"}, {"MCHAP_HIP_PIPE_PARTS": "16", "MCHAP_HIP_PIPE_FIRST": "2"},
                                   {"MCHAP_HIP_NO_BP_CACHE": "1"}, {"MCHAP_HIP_CACHE_SLOTS": "64"}],
                         ids=["default", "hand-over-after-1-step", "no-resume-rounds", "many-short-rounds", "16-lane-groups", "32-lane-groups",
                              "tables-completed-in-place", "tables-completed-by-16-waves", "no-base-product-cache", "tiny-llk-cache"])
def test_phased_sampler_hand_over_paths(sampler_kernel, monkeypatch, knobs):
    """The phased sampler (the library's default choice) against the speculative kernel on a batch with shallow reads,
    where chains keep moving: whatever the hand-over schedule -- chains handed over unsettled, handed back at once,
    resumed for a step at a time, or left to the final launch -- traces and llks are those of kernel 3, bit for bit."""
    if sampler_kernel != 5:
        pytest.skip("one pass is enough: the test picks its kernels itself")
    import ctypes as C

    from mchap_amd import DenovoMCMC, _lib
    from mchap_amd.synth import synth_units

    monkeypatch.delenv("MCHAP_HIP_KERNEL", raising=False)
    kw = dict(ploidy=4, n_alleles=[2] * 7, steps=300, chains=3, random_seed=11)
    shallow, _, _ = synth_units(24, ploidy=4, n_pos=7, n_reads=12, first_unit=5, window=(3, 7), qual=(3, 20))
    deep, _, _ = synth_units(24, ploidy=4, n_pos=7, n_reads=80, first_unit=900, window=(3, 7), qual=(30, 40))
    reads = list(shallow) + list(deep)  # (ragged: fit_batch pads the read axis)
    ref = DenovoMCMC(kernel=3, **kw).fit_batch(reads)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    if "MCHAP_HIP_PIPE_GROUP" in knobs:  # several chains per wavefront: instantiated in the parity suite's library only
        monkeypatch.setenv("MCHAP_HIP_TEST_KERNELS", "1")
    model = DenovoMCMC(kernel=0, **kw)
    got = model.fit_batch(reads)
    assert "phased" in model.last_sampler
    if "MCHAP_HIP_PIPE_GROUP" in knobs:
        assert ", %s, phased" % knobs["MCHAP_HIP_PIPE_GROUP"] in model.last_sampler
    moved = 0
    for a, b in zip(ref, got):
        assert np.array_equal(a.genotypes, b.genotypes)
        assert np.array_equal(a.llks, b.llks, equal_nan=True)
        moved += int((np.diff(a.llks[:, 50:], axis=1) != 0).any())
    assert moved > 0  # the batch does contain chains that still move late: the hand-back path ran


@pytest.mark.parametrize("n_reads", [1100, 1537, 2600, 4096])
def test_more_than_1024_reads(sampler_kernel, n_reads):
    """Read depths beyond 1024 rows per unit (up to 4096) run on the speculative and the phased sampler, whose
    likelihoods take the read chunks four at a time (kernel 4 shares that code); kernels 1 and 2 refuse them by name."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(2, ploidy=4, n_pos=6, n_reads=n_reads, first_unit=300 + n_reads, window=(2, 6), qual=(10, 40))
    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 6, steps=40, chains=2, random_seed=5, kernel=sampler_kernel)
    if sampler_kernel in (1, 2):
        with pytest.raises(NotImplementedError):
            model.fit_batch(list(reads))
        return
    try:
        _check(model, list(reads))
    except NotImplementedError:
        assert sampler_kernel == 4  # its LDS copy of the unit tables does not fit at every depth


@pytest.mark.parametrize("steps,chains", [(1, 2), (2, 1), (5, 3), (9, 2), (64, 5), (130, 40)])
def test_phased_sampler_short_runs_and_many_chains(sampler_kernel, monkeypatch, steps, chains):
    """Runs shorter than the phased sampler's first phase, runs that end inside a 64-step sweep of the coasting kernel,
    single chains and many chains per unit: same traces as kernel 3."""
    if sampler_kernel != 5:
        pytest.skip("one pass is enough: the test picks its kernels itself")
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    monkeypatch.delenv("MCHAP_HIP_KERNEL", raising=False)
    reads, _, _ = synth_units(5, ploidy=4, n_pos=8, n_reads=70, first_unit=41)
    kw = dict(ploidy=4, n_alleles=[2] * 8, steps=steps, chains=chains, random_seed=3)
    ref = DenovoMCMC(kernel=3, **kw).fit_batch(list(reads))
    got = DenovoMCMC(kernel=0, **kw).fit_batch(list(reads))
    for a, b in zip(ref, got):
        assert np.array_equal(a.genotypes, b.genotypes)
        assert np.array_equal(a.llks, b.llks, equal_nan=True)


def test_phased_sampler_equals_speculative_kernel_at_the_headline_shape(sampler_kernel, monkeypatch):
    """1024 loci of BASELINE.json's configuration (tetraploid, 8 SNVs, 200 reads, 1000 steps x 2 chains): the default
    (phased) sampler and kernel 3 write the same trace words, llks and status, bit for bit."""
    if sampler_kernel != 5:
        pytest.skip("one pass is enough: the test picks its kernels itself")
    import torch

    from mchap_amd import DenovoMCMC, _lib
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    monkeypatch.delenv("MCHAP_HIP_KERNEL", raising=False)
    reads, _, _ = synth_units(1024)
    out = {}
    for kernel in (3, 0):
        b = DenovoDeviceBatch(DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=1000, chains=2, random_seed=42, kernel=kernel), reads)
        b.run()
        torch.cuda.synchronize()
        out[kernel] = (b.d_trace.cpu().numpy(), b.d_llks.cpu().numpy(), b.d_status.cpu().numpy(), b.sampler_name)
    assert "phased" in out[0][3] and "phased" not in out[3][3]
    for a, c in zip(out[3][:3], out[0][:3]):
        assert np.array_equal(a, c)


@pytest.mark.parametrize("kernel", [2, 3, 5])
@pytest.mark.parametrize("shape", [(4, 8, 200), (4, 8, 100), (6, 10, 500), (4, 6, 150)])
def test_one_logarithm_per_group_of_reads(monkeypatch, kernel, shape):
    """Round 5: where the read weights are 0 / 1 a lane takes ONE logarithm per block of (up to four) read chunks -- the logarithm of
    the product of its reads' terms, mantissas multiplied and exponents summed (read_log.hpp read_log_product) -- instead of one
    per read.  The log likelihoods then differ from the per-read sums in the last bits only (every kernel forms the same groups,
    so the kernels stay bit-identical with each other: the other tests of this file); with flag 1048576 the kernels take a
    logarithm per read as before: the same genotypes at every step, log likelihoods equal to 1e-13, and the oracle's."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    K, M, R = shape
    monkeypatch.setenv("MCHAP_HIP_KERNEL", str(kernel))
    reads, _, _ = synth_units(6, ploidy=K, n_pos=M, n_reads=R, first_unit=900)
    kw = dict(ploidy=K, n_alleles=[2] * M, steps=150, chains=2, random_seed=31)
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    a = DenovoMCMC(**kw).fit_batch(list(reads))
    monkeypatch.setenv("MCHAP_HIP_FLAGS", str(1 << 20))
    b = DenovoMCMC(**kw).fit_batch(list(reads))
    differ = 0
    for x, y in zip(a, b):
        assert np.array_equal(x.genotypes, y.genotypes)
        np.testing.assert_allclose(x.llks, y.llks, rtol=1e-13)
        differ += int((x.llks != y.llks).sum())
    if R >= 200:
        assert differ > 0  # (the grouping is in use: some last bits do differ; a small unit whose chains hold one or two likelihoods may show none)
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    _check(DenovoMCMC(**kw), list(reads))


def test_weighted_reads_keep_a_logarithm_per_read(monkeypatch):
    """De-duplicated rows with counts: the weights are not 0 / 1, nothing is grouped -- the flag changes nothing at all."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import dedup_unit, synth_units

    reads, _, _ = synth_units(5, ploidy=4, n_pos=8, n_reads=200, first_unit=950, dedup=True)  # (phred ignored: equal rows do occur)
    pairs = [dedup_unit(r) for r in reads]
    assert all(int(p[1].max()) > 1 for p in pairs)
    kw = dict(ploidy=4, n_alleles=[2] * 8, steps=120, chains=2, random_seed=5)
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    a = DenovoMCMC(**kw).fit_batch([p[0] for p in pairs], [p[1] for p in pairs])
    monkeypatch.setenv("MCHAP_HIP_FLAGS", str(1 << 20))
    b = DenovoMCMC(**kw).fit_batch([p[0] for p in pairs], [p[1] for p in pairs])
    for x, y in zip(a, b):
        assert np.array_equal(x.genotypes, y.genotypes) and np.array_equal(x.llks, y.llks)
