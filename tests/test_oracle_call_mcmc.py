"""The oracle's restatement of `mchap call`'s sampler (calling/mcmc.py:15-453, calling/prior.py:30-113,
calling/classes.py:62-124) against vectors captured from the reference itself (tests/golden/call_mcmc.npz, made by
tests/golden/make_golden.py under the identity-njit shim): transition vectors of gibbs_options / mh_options, the greedy
initial genotype, and whole seeded CallingMCMC.fit traces reproduced step for step with the oracle's numpy-MT19937
generator.  CPU only."""
import json
import os

import numpy as np

from oracle import binding as orc


def _case(z, i):
    p = "c%d_" % i
    K, F, has_f = z[p + "meta"]
    K = int(K)
    prior = None if F < 0 else (float(F), z[p + "freqs"] if has_f else None)
    counts = z[p + "counts"]
    return p, K, prior, (None if counts.size == 0 else counts)


def test_transition_vectors_and_greedy_caller(golden_dir):
    z = np.load(os.path.join(golden_dir, "call_mcmc.npz"))
    for i in range(int(z["n_cases"])):
        p, K, prior, counts = _case(z, i)
        haps, reads = z[p + "haps"], z[p + "reads"]
        for state, vec in zip(z[p + "states"], z[p + "vectors"]):
            g, k = state[:K], int(state[K])
            for st in (0, 1):
                llks, lpri, probs = orc.call_step_options(reads, haps, g, k, st, counts, prior)
                np.testing.assert_allclose(llks, vec[st][0], rtol=1e-12)
                # a zero prior frequency: lgamma(0) = +inf under numba -> -inf / nan entries must agree too
                np.testing.assert_allclose(lpri, vec[st][1], rtol=1e-12, atol=1e-12, equal_nan=True)
                np.testing.assert_allclose(probs, vec[st][2], rtol=1e-10, atol=1e-300, equal_nan=True)
        assert orc.greedy_caller(reads, haps, K, counts, prior).tolist() == z[p + "greedy"].tolist()


def test_allele_prior_grid(golden_dir):
    with open(os.path.join(golden_dir, "call_mcmc_prior.json")) as f:
        grid = json.load(f)["grid"]
    assert len(grid) >= 30
    for c in grid:
        v = orc.log_genotype_allele_prior(c["g"], c["k"], c["H"], (c["F"], c["freqs"]))
        np.testing.assert_allclose(v, c["value"], rtol=1e-12)
        np.testing.assert_allclose(orc.log_genotype_allele_prior(c["g"], c["k"], c["H"], None), c["flat"], rtol=1e-15)


def test_seeded_fits_step_for_step(golden_dir):
    """CallingMCMC.fit with random_seed: the reference under the shim draws from numpy's legacy MT19937 (shuffle of the
    allele order, random_choice), which the oracle's ORC_RNG_NUMPY_MT19937 mode reproduces; the llk cache's
    first-ordering-wins behaviour (calling/likelihood.py:36-78) is part of what is pinned here."""
    z = np.load(os.path.join(golden_dir, "call_mcmc.npz"))
    n = int(z["n_cases"])
    assert n >= 6
    for i in range(n):
        p, K, prior, counts = _case(z, i)
        haps, reads = z[p + "haps"], z[p + "reads"]
        for st in (0, 1):
            g, l = orc.call_mcmc(reads, haps, K, steps=60, chains=2, step_type=st, read_counts=counts, prior=prior,
                                 rng_kind=orc.RNG_NUMPY_MT, seed=100 + i)
            assert np.array_equal(g, z[p + "trace%d_g" % st]), (i, st)
            np.testing.assert_allclose(l, z[p + "trace%d_l" % st], rtol=1e-10)
        g, l = orc.call_mcmc(reads, haps, K, steps=25, chains=1, step_type=0, read_counts=counts, prior=prior,
                             initial=z[p + "initial"], rng_kind=orc.RNG_NUMPY_MT, seed=7 + i)
        assert np.array_equal(g, z[p + "trace_ini_g"])
        np.testing.assert_allclose(l, z[p + "trace_ini_l"], rtol=1e-10)
