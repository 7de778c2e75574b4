"""GPU parity of the paths chains that keep moving take (round 4, late): the likelihood caches sized by the batch (mchap_hip.hip
cache_slots_of), the code table of a shallow unit in LDS up to 24 K rows (denovo_spec_kernel.hpp sct_off) and the two forms of
the coasting kernel (denovo_coast_kernel<1> / <COAST_NW_LIST>).  Traces bit for bit against the oracle on the same Philox
streams and against the same run with each piece switched off (tuning knobs); llks to 1e-10."""
import numpy as np
import pytest

from oracle import binding as orc
from tests.helpers import beta_break_table

pytestmark = pytest.mark.gpu

EVALUATE_BEHIND_KNOWN_MOVERS = 262144  # (flag 262144: spec_mutation evaluates every miss of a round's window, as before)
NO_CODED_TABLE = 4  # (flag 4: float64 rows from memory instead of the coded table -- and so no code table in LDS either)
ONE_WAVE_COASTING = 131072
NO_DECISION_CONTEXTS = 524288  # (flag 524288: the resumed chains of the phased sampler keep no decision contexts per genotype)


def _oracle(model, reads, counts, stream_id):
    M = reads.shape[1]
    cfg = orc.make_cfg(model.ploidy, model.steps, model.chains, model.inbreeding, model.temperatures, llk_cache_threshold=100,
                       rng_kind=orc.RNG_PHILOX, seed=model.random_seed, stream_id=stream_id,
                       break_table=beta_break_table(M, model.alpha, model.beta))
    g, l, code = orc.denovo_fit(cfg, reads, model.n_alleles, counts)
    assert code == 0
    return g, l


def _moves(tr):
    g = tr.genotypes
    return int((g[:, 1:] != g[:, :-1]).any(axis=(2, 3)).sum())


@pytest.mark.parametrize("kernel", [5, 3])
@pytest.mark.parametrize("shape", [(4, 8, 16, 2), (4, 8, 40, 2), (4, 13, 24, 2), (2, 10, 12, 2), (6, 7, 20, 2), (4, 6, 14, 3), (4, 21, 45, 2),
                                   (4, 23, 60, 2), (3, 20, 33, 3)])
def test_moving_chains_of_shallow_units(monkeypatch, kernel, shape):
    """Low-quality, shallow units: every chain keeps moving (hundreds of genotype changes) over few distinct genotypes -- the
    phase-ambiguous samples of real pileups.  Units of at most 64 reads evaluate their requests side by side from a code table in
    LDS of up to 24 K rows: (4, 13, 24) and up have more than the 24 rows haplotype 0's slots hold, (4, 23, 60) 46 rows at four
    reads per lane, (3, 20, 33, 3) 60 rows of a tri-allelic triploid.  (4, 21, 45): 84 bits per genotype -- the hashed-key cache.
    With the table in LDS, without any coded table (flag 4), and the oracle: the same traces."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    K, M, R, A = shape
    monkeypatch.setenv("MCHAP_HIP_KERNEL", str(kernel))
    reads, _, _ = synth_units(6, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, qual=(3, 20), window=(4, M) if M > 8 else (4, 8))
    kw = dict(ploidy=K, n_alleles=[A] * M, steps=400, chains=2, random_seed=7)
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    a = DenovoMCMC(**kw).fit_batch(list(reads))
    monkeypatch.setenv("MCHAP_HIP_FLAGS", str(NO_CODED_TABLE))
    b = DenovoMCMC(**kw).fit_batch(list(reads))
    assert sum(_moves(t) for t in a) > 6 * 2 * 100  # these chains do keep moving
    model = DenovoMCMC(**kw)
    for u, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x.genotypes, y.genotypes), "unit %d" % u
        np.testing.assert_allclose(x.llks, y.llks, rtol=1e-10, atol=1e-9)
        g, l = _oracle(model, reads[u], None, u)
        assert np.array_equal(x.genotypes, sort_haplotypes(g)), "unit %d" % u
        np.testing.assert_allclose(x.llks, l, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("kernel", [5, 3, 2])
@pytest.mark.parametrize("shape", [(4, 8, 16, None), (4, 13, 24, 0.2), (4, 21, 45, None), (2, 10, 12, 0.0), (6, 7, 20, None)])
def test_cutting_the_evaluations_behind_a_known_mover_is_results_neutral(monkeypatch, kernel, shape):
    """A chain with a history (more than 48 genotype changes) does not evaluate the proposals of a round that lie behind a
    sub-step whose cached proposal already says it moves (denovo_spec_kernel.hpp spec_mutation).  With the cut, without it
    (flag 262144) and on the lanes-over-chains kernel, which has no such thing: identical traces and log likelihoods."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    K, M, R, F = shape
    reads, _, _ = synth_units(6, ploidy=K, n_pos=M, n_reads=R, qual=(3, 20), window=(4, M) if M > 8 else (4, 8), first_unit=700)
    kw = dict(ploidy=K, n_alleles=[2] * M, inbreeding=F, steps=500, chains=2, random_seed=11)
    monkeypatch.setenv("MCHAP_HIP_KERNEL", str(kernel))
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    a = DenovoMCMC(**kw).fit_batch(list(reads))
    monkeypatch.setenv("MCHAP_HIP_FLAGS", str(EVALUATE_BEHIND_KNOWN_MOVERS))
    b = DenovoMCMC(**kw).fit_batch(list(reads))
    assert sum(_moves(t) for t in a) > 6 * 2 * 100
    for u, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x.genotypes, y.genotypes), "unit %d" % u
        assert np.array_equal(x.llks, y.llks)


def test_moving_chains_under_a_temperature_ladder_and_inbreeding(monkeypatch):
    from mchap_amd import DenovoMCMC
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(4, ploidy=4, n_pos=13, n_reads=20, qual=(3, 20), window=(4, 13), first_unit=9)
    kw = dict(ploidy=4, n_alleles=[2] * 13, inbreeding=0.1, steps=300, chains=2, temperatures=(0.2, 0.6, 1.0), random_seed=3)
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    a = DenovoMCMC(**kw).fit_batch(list(reads))
    model = DenovoMCMC(**kw)
    for u, x in enumerate(a):
        g, l = _oracle(model, reads[u], None, u)
        assert np.array_equal(x.genotypes, sort_haplotypes(g))
        np.testing.assert_allclose(x.llks, l, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("shape", [(4, 21, 45, 2), (4, 23, 80, 2), (6, 14, 30, 2), (8, 10, 24, 2), (3, 20, 33, 3)])
def test_moving_chains_with_wide_genotypes(monkeypatch, shape):
    """Genotypes of more than 63 bits (hashed cache tags verified against the stored words) whose chains never settle, through the
    phased sampler's hand-over rounds into its last launch, which runs them to the end: the oracle's traces.  (4, 23, 80): two
    read chunks -- no side-by-side evaluation; K = 6 and 8: the deep instantiation.)"""
    from mchap_amd import DenovoMCMC
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    K, M, R, A = shape
    monkeypatch.setenv("MCHAP_HIP_KERNEL", "5")
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    reads, _, _ = synth_units(5, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, qual=(3, 20), window=(4, M), first_unit=40)
    kw = dict(ploidy=K, n_alleles=[A] * M, steps=500, chains=2, random_seed=17)
    a = DenovoMCMC(**kw).fit_batch(list(reads))
    assert sum(_moves(t) for t in a) > 5 * 2 * 100
    model = DenovoMCMC(**kw)
    for u, x in enumerate(a):
        g, l = _oracle(model, reads[u], None, u)
        assert np.array_equal(x.genotypes, sort_haplotypes(g)), "unit %d" % u
        np.testing.assert_allclose(x.llks, l, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("slots", [64, 1024, 65536])
def test_cache_size_is_results_neutral(monkeypatch, slots):
    """The tables are sized by the batch unless MCHAP_HIP_CACHE_SLOTS names a size: any size gives the automatic size's traces."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(5, ploidy=4, n_pos=13, n_reads=24, qual=(3, 20), window=(4, 13), first_unit=31)
    kw = dict(ploidy=4, n_alleles=[2] * 13, steps=300, chains=2, random_seed=5)
    monkeypatch.delenv("MCHAP_HIP_CACHE_SLOTS", raising=False)
    a = DenovoMCMC(**kw).fit_batch(list(reads))
    monkeypatch.setenv("MCHAP_HIP_CACHE_SLOTS", str(slots))
    b = DenovoMCMC(**kw).fit_batch(list(reads))
    for x, y in zip(a, b):
        assert np.array_equal(x.genotypes, y.genotypes) and np.array_equal(x.llks, y.llks)


def test_coasting_forms_agree_and_long_lists_are_walked(monkeypatch):
    """The listed coasting launches run four wavefronts per chain over at most 1024 workgroups that take the list's entries in
    turn.  A batch whose chains are handed back by the thousand (shallow low-quality units among settled ones) exercises the
    walk; flag 131072 runs every coasting launch with one wavefront per chain: the same traces, and the oracle's."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    monkeypatch.setenv("MCHAP_HIP_KERNEL", "5")
    hot, _, _ = synth_units(1500, ploidy=4, n_pos=8, n_reads=40, qual=(3, 20), first_unit=100)
    cold, _, _ = synth_units(200, ploidy=4, n_pos=8, n_reads=40, first_unit=5000)
    reads = list(hot) + list(cold)
    kw = dict(ploidy=4, n_alleles=[2] * 8, steps=700, chains=2, random_seed=21)
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    a = DenovoMCMC(**kw).fit_batch(reads)
    monkeypatch.setenv("MCHAP_HIP_FLAGS", str(ONE_WAVE_COASTING))
    b = DenovoMCMC(**kw).fit_batch(reads)
    for u, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x.genotypes, y.genotypes), "unit %d" % u
        assert np.array_equal(x.llks, y.llks)
    model = DenovoMCMC(**kw)
    for u in list(range(0, 1500, 250)) + [1500, 1699]:
        g, l = _oracle(model, reads[u], None, u)
        assert np.array_equal(a[u].genotypes, sort_haplotypes(g)), "unit %d" % u
        np.testing.assert_allclose(a[u].llks, l, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("shape", [(4, 8, 40, None, 2), (4, 8, 16, None, 2), (4, 13, 24, 0.2, 2), (4, 21, 45, None, 2), (6, 7, 20, None, 2),
                                   (2, 10, 12, 0.0, 2), (4, 8, 100, None, 2), (8, 10, 24, None, 2), (3, 9, 33, None, 3)])
def test_decision_contexts_are_results_neutral(monkeypatch, shape):
    """Round 5: a chain with a history keeps a decision context per ordered genotype -- the move probability and successor
    likelihood of every sub-step, the interval totals -- in its region of the workspace (denovo_spec_kernel.hpp "decision
    contexts"): a move into a genotype it has held before is a context switch, not a round of proposals, probes and evaluations.
    What a context holds is what this arithmetic computed for that genotype, so the traces and likelihoods are those of the same
    run without contexts (flag 524288), bit for bit, and the oracle's.  (4, 8, 40): nearly every move finds its context; (4, 8,
    16): chains that wander -- their trial fails and they pause; (4, 21, 45): hashed tags, the slot's words decide; (4, 8, 100): two
    read chunks, no side-by-side evaluation; K = 8: the deep instantiation; (3, 9, 33, 3): tri-allelic -- no contexts at all.)"""
    from mchap_amd import DenovoMCMC
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    K, M, R, F, A = shape
    monkeypatch.setenv("MCHAP_HIP_KERNEL", "5")
    reads, _, _ = synth_units(6, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, qual=(3, 20), window=(4, M) if M > 8 else (4, 8), first_unit=300)
    kw = dict(ploidy=K, n_alleles=[A] * M, inbreeding=F, steps=1500, chains=2, random_seed=23)
    monkeypatch.delenv("MCHAP_HIP_FLAGS", raising=False)
    a = DenovoMCMC(**kw).fit_batch(list(reads))
    monkeypatch.setenv("MCHAP_HIP_FLAGS", str(NO_DECISION_CONTEXTS))
    b = DenovoMCMC(**kw).fit_batch(list(reads))
    assert sum(_moves(t) for t in a) > 6 * 2 * 30  # (every chain passes the threshold of MCHAP_CTX_GEN genotype changes)
    model = DenovoMCMC(**kw)
    for u, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x.genotypes, y.genotypes), "unit %d" % u
        assert np.array_equal(x.llks, y.llks)
    for u in (0, 3):
        g, l = _oracle(model, reads[u], None, u)
        assert np.array_equal(a[u].genotypes, sort_haplotypes(g)), "unit %d" % u
        np.testing.assert_allclose(a[u].llks, l, rtol=1e-10, atol=1e-9)
