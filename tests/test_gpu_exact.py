"""GPU parity of the exact caller against vectors captured from the reference (tests/golden/exact.npz) and
against the CPU oracle on larger shapes.  Tolerances: fp64 path 1e-9 relative; float32 path as stored float32
(<= 1 ulp of float32 on the likelihoods, 3e-5 relative on posteriors formed with float32 arithmetic)."""
import os

import numpy as np
import pytest

from oracle import binding as orc

pytestmark = pytest.mark.gpu


def _rc(a):
    return None if a.size == 0 else a


def test_against_reference_goldens(golden_dir):
    from mchap_amd import calling

    z = np.load(os.path.join(golden_dir, "exact.npz"))
    for i in range(int(z["n_cases"])):
        p = "c%d_" % i
        reads, haps, rc = z[p + "reads"], z[p + "haps"], _rc(z[p + "counts"])
        K, F = z[p + "meta"]
        K = int(K)
        fr = z[p + "freqs_prior"]
        prior = None if F < 0 else (float(F), None if fr.size == 0 else fr)
        llks = calling.genotype_likelihoods(reads, K, haps, rc)
        assert llks.dtype == np.float32
        np.testing.assert_allclose(llks, z[p + "llks_f32"], rtol=2.5e-7)
        post = calling.genotype_posteriors(z[p + "llks_f32"], K, len(haps), prior)
        assert post.dtype == np.float64
        np.testing.assert_allclose(post, z[p + "post_from_f32"], rtol=3e-5, atol=1e-10)
        post64 = calling.genotype_posteriors(z[p + "llks_f32"].astype(np.float64), K, len(haps), prior)
        np.testing.assert_allclose(post64, z[p + "post_from_f64"], rtol=1e-9, atol=1e-300)
        f, c, o = calling.posterior_allele_frequencies(z[p + "post_from_f32"], K, len(haps))
        np.testing.assert_allclose(np.stack([f, c, o]), z[p + "afreq"], rtol=1e-12)
        mode = calling.posterior_mode(reads, K, haps, rc, prior, True, True, True)
        assert mode[0].tolist() == z[p + "mode_alleles"].tolist()
        np.testing.assert_allclose(mode[1:4], z[p + "mode_stats"], rtol=1e-9)
        np.testing.assert_allclose(np.stack([mode[4], mode[5]]), z[p + "mode_freqs"], rtol=1e-9, atol=1e-300)
        ag, ap = calling.alternate_dosage_posteriors(mode[0], post)
        assert np.array_equal(ag, z[p + "alt_genotypes"])
        np.testing.assert_allclose(ap, z[p + "alt_probs"], rtol=3e-5, atol=1e-10)
        # index helpers
        idx = calling.genotype_alleles_as_index(mode[0])
        assert calling.index_as_genotype_alleles(idx, K).tolist() == mode[0].tolist()


@pytest.mark.parametrize("K,H,M,R,U", [(4, 8, 8, 120, 3), (6, 10, 10, 200, 2), (2, 30, 6, 64, 2), (8, 5, 5, 40, 2),
                                       (4, 24, 6, 1500, 2),  # 288 KB of products -- the passes tile the reads
                                       (10, 6, 8, 90, 2), (12, 5, 7, 130, 2), (15, 4, 6, 60, 2), (9, 6, 100, 70, 2)],
                         # (round 5: ploidies 9 to 15 -- what the general de novo sampler assembles --, and haplotypes of 100 SNVs)
                         ids=["K4", "K6", "K2-H30", "K8", "tiled-reads", "K10", "K12", "K15", "K9-100snvs"])
def test_batch_against_oracle(K, H, M, R, U):
    from mchap_amd import calling
    from mchap_amd.synth import synth_units

    rng = np.random.default_rng(K * 100 + H)
    reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=H, window=(3, M))
    haps = np.zeros((U, H, M), np.int8)
    for u in range(U):
        pool = np.unique(np.concatenate([truth[u], rng.integers(0, 2, size=(4 * H, M)).astype(np.int8)]), axis=0)
        rng.shuffle(pool)
        haps[u] = pool[:H]
    counts = rng.integers(1, 4, size=(U, R)).astype(np.int64)
    F = np.array([0.1, 0.4, 0.0][:U])
    fr = rng.dirichlet(np.ones(H), size=U)
    for prior in (None, (F, None), (F, fr)):
        out = calling.posterior_mode_batch(reads, K, haps, counts, prior, True, True, True)
        for u in range(U):
            pr = None if prior is None else (float(F[u]), None if prior[1] is None else fr[u])
            a, ml, mp, sp, fq, oc = orc.posterior_mode(reads[u], K, haps[u], counts[u], pr)
            assert out[0][u].tolist() == a.tolist()
            np.testing.assert_allclose([out[1][u], out[2][u], out[3][u]], [ml, mp, sp], rtol=1e-9)
            np.testing.assert_allclose(out[4][u], fq, rtol=1e-9, atol=1e-300)
            np.testing.assert_allclose(out[5][u], oc, rtol=1e-9, atol=1e-300)
    l32 = calling.genotype_likelihoods(reads[0], K, haps[0], counts[0])
    e32, e64 = orc.genotype_likelihoods(reads[0], K, haps[0], counts[0])
    np.testing.assert_allclose(l32, e32, rtol=2.5e-7)


def test_zero_frequency_allele_and_zero_reads():
    from mchap_amd import calling
    from mchap_amd.synth import synth_units

    reads, _, truth = synth_units(1, ploidy=4, n_pos=5, n_reads=30)
    haps = np.unique(np.concatenate([truth[0], np.zeros((1, 5), np.int8), np.ones((1, 5), np.int8)]), axis=0)
    H = len(haps)
    fr = np.full(H, 1.0 / (H - 1))
    fr[0] = 0.0  # lgamma(0) = +inf under numba: every genotype carrying allele 0 gets zero posterior
    out = calling.posterior_mode(reads[0], 4, haps, None, (0.2, fr), True, True, True)
    assert out[4][0] == 0.0 and out[5][0] == 0.0
    assert np.isclose(out[4].sum(), 1.0)
    # no reads: the posterior is the prior
    empty = np.empty((0, 5, 2))
    llks = calling.genotype_likelihoods(empty, 4, haps)
    assert np.all(llks == 0)
    post = calling.genotype_posteriors(llks.astype(np.float64), 4, H, None)
    np.testing.assert_allclose(post, 1.0 / len(post), rtol=1e-12)


@pytest.mark.parametrize("K,H,M,R,U", [(4, 6, 6, 60, 5), (6, 7, 5, 33, 3), (2, 12, 6, 64, 4), (3, 20, 6, 1100, 2), (1, 6, 5, 40, 3), (8, 5, 6, 50, 2)],
                         ids=["K4", "K6", "K2", "tiled-reads", "haploid", "K8"])
def test_device_batch_arrays_and_streaming_against_oracle(K, H, M, R, U):
    """mchap_exact_call_batch_device: every output of the streaming form and of the array form (likelihoods, posteriors
    and the summaries call_exact.py:126-159 derives from the posterior array) for a batch, against the oracle unit by
    unit; the streaming form forms no per-genotype array (two passes), the array form needs no workspace."""
    from mchap_amd import calling
    from mchap_amd.device import ExactDeviceBatch
    from mchap_amd.synth import synth_units

    rng = np.random.default_rng(K * 1000 + H)
    reads, _, truth = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=3 * H, window=(2, M), qual=(2, 12))
    haps = np.zeros((U, H, M), np.int8)
    for u in range(U):
        pool = np.unique(rng.integers(0, 2, size=(6 * H, M)).astype(np.int8), axis=0)
        rng.shuffle(pool)
        haps[u] = pool[:H]
    counts = rng.integers(1, 4, size=(U, R)).astype(np.int64)
    F = rng.choice([0.0, 0.1, 0.3], size=U)
    fr = rng.dirichlet(np.ones(H), size=U)
    for prior in (None, (F, None), (F, fr)):
        batch = ExactDeviceBatch(reads, K, haps, counts, prior)
        batch.run(streaming=True, arrays=True, llks64=True)
        mode = batch.mode_results()
        arr = batch.array_results()
        # second pass from the joint log-probabilities kept in the (larger) workspace == second pass from scratch
        plain = ExactDeviceBatch(reads, K, haps, counts, prior, cache_joint=False)
        assert plain.ws_bytes < batch.ws_bytes
        plain.run(streaming=True, arrays=False)
        for a, b in zip(mode, plain.mode_results()):
            assert np.array_equal(a, b, equal_nan=True)
        for u in range(U):
            pr = None if prior is None else (float(F[u]), None if prior[1] is None else fr[u])
            a, ml, mp, sp, fq, oc = orc.posterior_mode(reads[u], K, haps[u], counts[u], pr)
            assert mode[0][u].tolist() == a.tolist()
            np.testing.assert_allclose([mode[1][u], mode[2][u], mode[3][u]], [ml, mp, sp], rtol=1e-9)
            np.testing.assert_allclose(mode[4][u], fq, rtol=1e-9, atol=1e-300)
            np.testing.assert_allclose(mode[5][u], oc, rtol=1e-9, atol=1e-300)
            e32, e64 = orc.genotype_likelihoods(reads[u], K, haps[u], counts[u])
            np.testing.assert_allclose(arr["llks"][u], e32, rtol=2.5e-7)
            np.testing.assert_allclose(batch._host("llks64", (U, batch.G))[u], e64, rtol=1e-10)
            ref = orc.genotype_posteriors(e32, K, H, pr)
            ulp = float(np.spacing(np.float32(np.abs(e32).max())))
            np.testing.assert_allclose(arr["posteriors"][u], ref, rtol=max(3e-5, 4 * ulp), atol=1e-12)
            # summaries of the (device's own) posterior array: exact sums over the array
            post = arr["posteriors"][u]
            idx = int(np.argmax(post))
            assert arr["alleles"][u].tolist() == calling.index_as_genotype_alleles(idx, K).tolist()
            assert arr["prob"][u] == post[idx]
            rf, rc_, ro = orc.posterior_allele_frequencies(post, K, H)
            np.testing.assert_allclose(np.stack([arr["freqs"][u], arr["counts"][u], arr["occur"][u]]), np.stack([rf, rc_, ro]), rtol=1e-9, atol=1e-300)
            ag, ap = calling.alternate_dosage_posteriors(arr["alleles"][u], post)
            np.testing.assert_allclose(arr["support_prob"][u], ap.sum(), rtol=1e-12)
    # the convenience wrapper
    res = calling.call_arrays_batch(reads, K, haps, counts, (F, fr))
    np.testing.assert_allclose(res["posteriors"], arr["posteriors"], rtol=0, atol=0)
