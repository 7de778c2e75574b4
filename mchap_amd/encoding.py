"""Input encoders that define the read tensor the kernels consume.

Mirrors the reference's mchap/encoding/integer/transcode.py:16-77 (`as_probabilistic`),
mchap/io/util.py:40-53 (`prob_of_qual`) and the de-duplication of application/baseclass.py:207
(mset.unique_counts), with the same names and argument meaning.
"""
import numpy as np

from .synth import PFEIFFER_ERROR, dedup_unit  # noqa: F401

__all__ = ["as_probabilistic", "prob_of_qual", "encode_read_distributions", "unique_counts"]


def as_probabilistic(array, n_alleles=4, p=1.0, error_factor=3, dtype=float):
    """Integer encoded alleles -> probabilistic row vectors (same name and arguments as the reference's encoder).

    Rules: the called allele gets `p`; every other allele gets `(1 - p) / error_factor`
    (error_factor stays 3 whatever n_alleles is); a gap (call < 0) makes the whole position NaN; alleles the position
    does not have (a >= n_alleles[j]) are zero -- applied last, so a gap at a biallelic position of a 3-allele tensor
    reads [nan, nan, 0]."""
    calls = np.asarray(array)
    if calls.shape[-1] == 0:
        return np.empty(calls.shape + (0,), dtype=dtype)
    n_alleles = np.broadcast_to(np.asarray(n_alleles), calls.shape)
    p = np.broadcast_to(np.asarray(p, dtype=float), calls.shape)
    share = np.broadcast_to((1.0 - p) / np.asarray(error_factor, dtype=float), calls.shape)
    allele = np.arange(int(np.max(n_alleles)))
    called = calls[..., None] == allele
    out = np.where(called, p[..., None], share[..., None])
    out = np.where((calls < 0)[..., None], np.nan, out)
    out = np.where(allele >= n_alleles[..., None], 0.0, out)
    return out.astype(dtype, copy=False)


def prob_of_qual(qual):
    """Phred quality -> probability that the call is correct (reference io/util.py:40-53)."""
    return 1 - (10 ** (np.asarray(qual) / -10))


def encode_read_distributions(n_alleles, read_calls, read_quals=None, error_rate=PFEIFFER_ERROR, gaps=True):
    """int8 calls [R, M] (+ optional phred quals) -> float64 tensor [R, M, A] (reference io/bam.py:251-289):
    p = (1 - error_rate) * prob_of_qual(q), or 1 - error_rate when quals are ignored (the CLI default)."""
    read_calls = np.asarray(read_calls)
    n_pos = len(n_alleles)
    max_allele = int(np.max(n_alleles)) if n_pos else 0
    n_reads = len(read_calls)
    if n_reads == 0 or n_pos == 0:
        return np.empty((n_reads, n_pos, max_allele), dtype=float)
    if read_quals is None:
        probs = 1.0 - error_rate
    else:
        probs = prob_of_qual(read_quals) * (1.0 - error_rate)
    dists = as_probabilistic(read_calls, n_alleles, probs)
    if gaps is False:
        dists[np.isnan(dists)] = 0
    return dists


def unique_counts(reads):
    """Distinct read rows in order of first appearance and their counts (reference mset.unique_counts as used at
    application/baseclass.py:207)."""
    return dedup_unit(np.asarray(reads))
