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
