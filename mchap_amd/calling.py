"""Exact genotype caller on MI355X HIP kernels: drop-in for the module-level functions of the reference's
mchap/calling/exact.py (posterior_mode 156-249, genotype_likelihoods 266-292, genotype_posteriors 295-329,
posterior_allele_frequencies 332-369, alternate_dosage_posteriors 372-407) -- same names, arguments and results.
`posterior_mode_batch` runs many units that share a shape in one launch.  No CPU fallback.
"""
import ctypes as C
from itertools import combinations_with_replacement
from math import comb

import numpy as np

from . import _lib

__all__ = [
    "posterior_mode",
    "posterior_mode_batch",
    "genotype_likelihoods",
    "genotype_posteriors",
    "posterior_allele_frequencies",
    "alternate_dosage_posteriors",
    "count_unique_genotypes",
    "genotypes_in_vcf_order",
    "call_arrays_batch",
    "genotype_alleles_as_index",
    "index_as_genotype_alleles",
]


def count_unique_genotypes(u_haps, ploidy):
    """Number of genotypes of `ploidy` over `u_haps` haplotypes (reference combinatorics.py:35-54)."""
    return comb(u_haps + ploidy - 1, ploidy) if u_haps > 0 else 0


def _cwr(n, k):
    return comb(n + k - 1, k) if n > 0 else 0


def genotype_alleles_as_index(alleles):
    """VCF order index of ascending alleles (reference jitutils.py:253-276)."""
    index = 0
    for i, a in enumerate(alleles):
        if a < 0:
            raise ValueError("Allele numbers must be >= 0.")
        index += _cwr(int(a), i + 1)
    return index


def index_as_genotype_alleles(index, ploidy):
    """Alleles of the genotype at a VCF order index (reference jitutils.py:279-318)."""
    out = np.full(ploidy, -2, np.int64)
    if index < 0:
        out[:] = -1
        return out
    remainder = index
    for p in range(ploidy, 0, -1):
        n, new, prev = -1, 0, 0
        while new <= remainder:
            n += 1
            prev = new
            new = _cwr(n, p)
        remainder -= prev
        out[p - 1] = n - 1
    return out


def _reads(reads, read_counts):
    reads = np.ascontiguousarray(reads, dtype=np.float64)
    n_reads = reads.shape[0]
    if n_reads == 0:
        # a missing sample is one all-NaN read (reference assemble/mcmc.py:132-137, snpcalling.py:42-46)
        reads = np.full((1,) + reads.shape[1:], np.nan)
        read_counts = None
    rc = None if read_counts is None else np.ascontiguousarray(read_counts, dtype=np.int64)
    return reads, rc


def genotype_likelihoods(reads, ploidy, haplotypes, read_counts=None):
    """Log likelihood of every genotype in VCF order, float32 as the reference stores it."""
    reads, rc = _reads(reads, read_counts)
    haps = np.ascontiguousarray(haplotypes, dtype=np.int8)
    R, M, A = reads.shape
    G = count_unique_genotypes(len(haps), ploidy)
    out = np.full(G, np.nan, np.float32)
    _lib.check(_lib.lib().mchap_exact_genotype_likelihoods(
        _lib.ptr(reads), R, M, A, _lib.ptr(rc), _lib.ptr(haps), len(haps), int(ploidy), _lib.ptr(out), None))
    return out


def genotype_posteriors(log_likelihoods, ploidy, n_alleles, prior=None):
    """Posterior probability of every genotype in VCF order (float64 array; float32 arithmetic when the
    likelihoods are float32, as in the reference)."""
    llks = np.ascontiguousarray(log_likelihoods)
    if llks.dtype not in (np.float32, np.float64):
        llks = llks.astype(np.float64)
    has = 0 if prior is None else 1
    F = 0.0 if prior is None else float(prior[0])
    fr = None if (prior is None or prior[1] is None) else np.ascontiguousarray(prior[1], dtype=np.float64)
    out = np.zeros(len(llks), dtype=np.float64)
    _lib.check(_lib.lib().mchap_exact_genotype_posteriors(
        _lib.ptr(llks), int(llks.dtype == np.float32), C.c_int64(len(llks)), int(ploidy), int(n_alleles), has,
        C.c_double(F), _lib.ptr(fr), _lib.ptr(out)))
    return out


def posterior_allele_frequencies(posteriors, ploidy, n_alleles):
    """(mean allele frequencies, allele counts, occurrence probabilities) of a posterior over VCF ordered genotypes
    (one pass of the device kernel over the array)."""
    post = np.ascontiguousarray(posteriors, dtype=np.float64)
    freqs, counts, occur = np.zeros(n_alleles), np.zeros(n_alleles), np.zeros(n_alleles)
    _lib.check(_lib.lib().mchap_exact_posterior_summaries(
        _lib.ptr(post), C.c_int64(len(post)), int(ploidy), int(n_alleles), None, None, None, _lib.ptr(freqs), _lib.ptr(counts),
        _lib.ptr(occur)))
    return freqs, counts, occur


def genotypes_in_vcf_order(n, ploidy):
    """Alleles [n, ploidy] of the first n genotypes in VCF order: the combinatorial number system read from the
    highest position down (what index_as_genotype_alleles does one index at a time), vectorised over the indices."""
    rem = np.arange(n, dtype=np.int64)
    out = np.zeros((n, ploidy), dtype=np.int64)
    top = 1
    while _cwr(top, ploidy) <= max(n - 1, 0):
        top += 1
    for p in range(ploidy, 0, -1):
        table = np.array([_cwr(a, p) for a in range(top + 2)], dtype=np.int64)  # genotypes of p alleles below allele a
        a = np.searchsorted(table, rem, side="right") - 1
        out[:, p - 1] = a
        rem = rem - table[a]
    return out


def alternate_dosage_posteriors(genotype_alleles, probabilities):
    """Posterior of every dosage variant of the genotype's support (calling/exact.py:363-407): the genotypes that hold each allele
    of the support at least once, in VCF order, with their entries of `probabilities`.  A variant is the support plus a multiset of
    `ploidy - len(support)` extra copies; all of them are formed as one array (multisets as rows of indices into the support), their
    VCF indices by the combinatorial number system over the sorted rows, and the rows ordered by index."""
    support = np.unique(np.asarray(genotype_alleles))
    ploidy, extra = len(genotype_alleles), len(genotype_alleles) - len(support)
    combos = list(combinations_with_replacement(range(len(support)), extra))  # (extra == 0: the one empty multiset)
    picks = np.array(combos, dtype=np.int64).reshape(len(combos), extra)
    rows = np.sort(np.concatenate([np.broadcast_to(support, (len(picks), len(support))), support[picks]], axis=1), axis=1).astype(np.int64)
    # index of a sorted row a_1 <= ... <= a_K in VCF order: sum_k C(a_k + k - 1, k)
    index = np.zeros(len(rows), dtype=np.int64)
    for k in range(1, ploidy + 1):
        index += np.array([comb(int(a) + k - 1, k) for a in rows[:, k - 1]], dtype=np.int64)
    order = np.argsort(index, kind="stable")
    return rows[order], np.asarray(probabilities, dtype=float)[index[order]]


def posterior_mode_batch(reads, ploidy, haplotypes, read_counts=None, prior=None, return_support_prob=False,
                         return_posterior_frequencies=False, return_posterior_occurrence=False):
    """posterior_mode for a batch: reads [U, R, M, A], haplotypes [U, H, M] (or [H, M] shared),
    read_counts [U, R] or None, prior = None | (inbreeding scalar or [U], frequencies None | [H] | [U, H])."""
    reads = np.ascontiguousarray(reads, dtype=np.float64)
    U, R, M, A = reads.shape
    haps = np.asarray(haplotypes, dtype=np.int8)
    if haps.ndim == 2:
        haps = np.broadcast_to(haps, (U,) + haps.shape)
    haps = np.ascontiguousarray(haps)
    H = haps.shape[1]
    rc = None if read_counts is None else np.ascontiguousarray(read_counts, dtype=np.int64)
    has = 0 if prior is None else 1
    F = fr = None
    if prior is not None:
        F = np.ascontiguousarray(np.broadcast_to(np.asarray(prior[0], dtype=np.float64), (U,)))
        if prior[1] is not None:
            fr = np.ascontiguousarray(np.broadcast_to(np.asarray(prior[1], dtype=np.float64), (U, H)))
    K = int(ploidy)
    alleles = np.zeros((U, K), np.int64)
    mllk, mprob = np.zeros(U), np.zeros(U)
    sprob = np.zeros(U) if return_support_prob else None
    want_f = return_posterior_frequencies or return_posterior_occurrence
    freqs = np.zeros((U, H)) if want_f else None
    occur = np.zeros((U, H)) if want_f else None
    _lib.check(_lib.lib().mchap_exact_posterior_mode_batch(
        U, _lib.ptr(reads), R, M, A, _lib.ptr(rc), _lib.ptr(haps), H, K, has, _lib.ptr(F), _lib.ptr(fr),
        _lib.ptr(alleles), _lib.ptr(mllk), _lib.ptr(mprob), _lib.ptr(sprob), _lib.ptr(freqs), _lib.ptr(occur)))
    result = [alleles, mllk, mprob]
    if return_support_prob:
        result.append(sprob)
    if return_posterior_frequencies:
        result.append(freqs)
    if return_posterior_occurrence:
        result.append(occur)
    return tuple(result)


def posterior_mode(reads, ploidy, haplotypes, read_counts=None, prior=None, return_support_prob=False,
                   return_posterior_frequencies=False, return_posterior_occurrence=False):
    """Posterior mode genotype with statistics from a set of known haplotypes (streaming form: no array over
    all genotypes is returned).  Returns (mode_alleles, mode_llk, mode_probability[, mode_support_probability]
    [, mean_allele_frequencies][, allele_occurrence_probability])."""
    reads, rc = _reads(reads, read_counts)
    out = posterior_mode_batch(reads[None], ploidy, np.asarray(haplotypes)[None], None if rc is None else rc[None], prior,
                               return_support_prob, return_posterior_frequencies, return_posterior_occurrence)
    res = [out[0][0], float(out[1][0]), float(out[2][0])]
    k = 3
    if return_support_prob:
        res.append(float(out[k][0]))
        k += 1
    if return_posterior_frequencies:
        res.append(out[k][0])
        k += 1
    if return_posterior_occurrence:
        res.append(out[k][0])
    return tuple(res)


def call_arrays_batch(reads, ploidy, haplotypes, read_counts=None, prior=None, return_arrays=True):
    """The array form of the exact caller for a batch (what `mchap call-exact --report GL GP` computes per sample,
    reference application/call_exact.py:126-159): genotype_likelihoods (float32) -> genotype_posteriors -> the
    posterior array's mode, the probability of the mode's support, posterior_allele_frequencies -- one device call,
    no host loop over genotypes or units.

    reads [U, R, M, A], haplotypes [U, H, M] or [H, M], read_counts [U, R] or None,
    prior = None | (inbreeding scalar or [U], frequencies None | [H] | [U, H]).
    Returns a dict: alleles [U, K], prob [U], support_prob [U], freqs / counts / occur [U, H] and, with
    return_arrays, llks float32 [U, G] and posteriors float64 [U, G]."""
    from .device import ExactDeviceBatch

    batch = ExactDeviceBatch(reads, ploidy, haplotypes, read_counts, prior)
    batch.run(arrays=True, streaming=False)
    return batch.array_results(return_arrays)
