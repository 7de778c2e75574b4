"""Shared helpers for the tests (not part of the product)."""
import numpy as np
from scipy import stats as _stats


def beta_break_table(n_pos, alpha=1.0, beta=3.0):
    """Row m = Beta(alpha, beta) CDF increments over m het bases (what DenovoMCMC computes with scipy,
    reference assemble/mcmc.py:429-452), zero padded to n_pos columns."""
    tab = np.zeros((n_pos + 1, max(n_pos, 1)))
    dist = _stats.beta(alpha, beta)
    for m in range(1, n_pos + 1):
        pts = np.arange(1, m + 1) / m
        probs = dist.cdf(pts)
        probs[1:] = probs[1:] - probs[:-1]
        tab[m, :m] = probs
    return tab


def lexsort_rows(g):
    """Canonical haplotype order of one genotype [K, M] (position 0 most significant)."""
    g = np.asarray(g)
    return g[np.lexsort(np.flip(g, axis=-1).T)]


def assert_same_posterior(got_genotypes, got_probs, exp_genotypes, exp_probs, rtol=1e-15):
    """Posterior lists (probability descending) are equal: the same probabilities in the same order, and the same
    genotypes -- in the same order where the probabilities differ, as the same SET inside a run of tied probabilities.
    The reference orders with np.flip(np.argsort(probs)) (assemble/classes.py:316-325); numpy's default argsort is not
    stable (the AVX-512 sort of numpy 2.x reorders ties even among 4 elements), so the order inside a tie is not
    defined by the reference.  This build's rule (descending first appearance) is pinned separately."""
    got_genotypes, exp_genotypes = np.asarray(got_genotypes), np.asarray(exp_genotypes)
    got_probs, exp_probs = np.asarray(got_probs, float), np.asarray(exp_probs, float)
    assert got_genotypes.shape == exp_genotypes.shape
    np.testing.assert_allclose(got_probs, exp_probs, rtol=rtol)
    n = len(exp_probs)
    i = 0
    while i < n:
        j = i
        while j < n and exp_probs[j] == exp_probs[i]:
            j += 1
        a = sorted(g.tobytes() for g in got_genotypes[i:j].astype(np.int8))
        b = sorted(g.tobytes() for g in exp_genotypes[i:j].astype(np.int8))
        assert a == b, (i, j)
        i = j
