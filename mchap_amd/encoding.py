"""Input encoders that define the read tensor the kernels consume.

Mirrors the reference's mchap/encoding/integer/transcode.py:16-77 (`as_probabilistic`),
mchap/io/util.py:40-53 (`prob_of_qual`) and the de-duplication of application/baseclass.py:207
(mset.unique_counts), with the same names and argument meaning.
"""
import numpy as np

from .synth import PFEIFFER_ERROR, dedup_unit  # noqa: F401

__all__ = ["as_probabilistic", "prob_of_qual", "encode_read_distributions", "unique_counts"]


def as_probabilistic(array, n_alleles=4, p=1.0, error_factor=3, dtype=float):
    """Integer encoded alleles -> probabilistic row vectors.

    The called allele gets `p`, every other allele `(1 - p) / error_factor` (error_factor stays 3 whatever
    n_alleles is), alleles >= n_alleles[j] are zeroed, and a gap (call < 0) makes the whole position NaN
    *before* that zero mask -- so a gap at a biallelic position of a 3-allele tensor is [nan, nan, 0]."""
    array = np.asarray(array)
    n_alleles = np.asarray(n_alleles)
    error_factor = np.asarray(error_factor)
    p = np.asarray(p)
    if array.shape[-1] == 0:
        return np.empty(array.shape + (0,), dtype=dtype)
    alleles = np.arange(np.max(n_alleles))
    onehot = array[..., None] == alleles
    new = ((1 - p) / error_factor)[..., None] * ~onehot
    calls = p[..., None] * onehot
    new[onehot] = calls[onehot]
    new[array < 0] = np.nan
    new[..., n_alleles[..., None] <= alleles] = 0
    return new.astype(dtype, copy=False)


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
