"""Trace and posterior containers with the reference's API.

Mirrors mchap/assemble/classes.py of the reference (Assembler 16-52,
PosteriorGenotypeDistribution 55-166, GenotypeSupportDistribution 169-244,
GenotypeMultiTrace 247-376) -- same names, arguments and results -- implemented with
vectorised numpy over packed haplotype keys instead of per-step Python loops.  Traces
produced by the HIP sampler arrive already in canonical haplotype order, so no per-step
sort runs on the host.
"""
from dataclasses import dataclass
from functools import reduce

import numpy as np

__all__ = [
    "Assembler",
    "PosteriorGenotypeDistribution",
    "GenotypeSupportDistribution",
    "GenotypeMultiTrace",
]


def _row_keys(array):
    """One bytes key per element of the outer dimension (the reference keys on tobytes())."""
    a = np.ascontiguousarray(array)
    n = len(a)
    if n == 0:
        return np.zeros(0, dtype="V1")
    width = a.dtype.itemsize * int(np.prod(a.shape[1:], dtype=np.int64))
    if width == 0:
        return np.zeros(n, dtype="V1")
    return a.reshape(n, -1).view("V%d" % width).reshape(n)


def _first_occurrence_unique(array):
    """(index of first occurrence of each distinct element, in order of appearance; inverse)."""
    keys = _row_keys(array)
    _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    return first[order], rank[inverse.reshape(-1)]


def unique_rows(array):
    """Distinct elements of the outer dimension in order of first appearance (reference mset.unique)."""
    idx, _ = _first_occurrence_unique(array)
    return array[idx]


def unique_counts(array):
    """Distinct elements in order of first appearance and their counts (reference mset.unique_counts)."""
    idx, inv = _first_occurrence_unique(array)
    return array[idx], np.bincount(inv, minlength=len(idx))


def sort_haplotypes(genotypes):
    """Canonical haplotype order of every genotype [..., K, M]: lexicographic, position 0 most significant
    (reference encoding/integer/sequence.py:78-110 applied per step in classes.py:275-278)."""
    g = np.asarray(genotypes)
    K, M = g.shape[-2:]
    flat = g.reshape(-1, K, M)
    if M == 0 or len(flat) == 0:
        return g.copy()
    # stable sorts from the least to the most significant position == np.lexsort per genotype
    order = np.tile(np.arange(K), (len(flat), 1))
    rows = np.arange(len(flat))[:, None]
    for j in range(M - 1, -1, -1):
        col = flat[rows, order, j]
        o = np.argsort(col, axis=1, kind="stable")
        order = np.take_along_axis(order, o, axis=1)
    return flat[rows, order].reshape(g.shape)


@dataclass
class Assembler(object):
    """Abstract base class for haplotype assemblers (reference classes.py:16-52)."""

    @classmethod
    def parameterize(cls, *args, **kwargs):
        return cls(*args, **kwargs)

    def fit(self):
        raise NotImplementedError()


@dataclass
class PosteriorGenotypeDistribution(object):
    """Posterior distribution over (finite and countable) genotypes.

    genotypes : int [n_genotypes, ploidy, n_positions]; probabilities : float [n_genotypes]."""

    genotypes: np.ndarray
    probabilities: np.ndarray

    def mode(self):
        idx = np.argmax(self.probabilities)
        return self.genotypes[idx], self.probabilities[idx]

    def _support_labels(self):
        """label[i] = index of the first genotype with the same set of unique haplotypes."""
        n = len(self.genotypes)
        sig = []
        for gen in self.genotypes:
            sig.append(unique_rows(gen).tobytes())
        labels = np.zeros(n, dtype=int)
        seen = {}
        for i, s in enumerate(sig):
            labels[i] = seen.setdefault(s, i)
        return labels

    def _support_sums(self):
        labels = self._support_labels()
        firsts = np.unique(labels)  # ascending == order of first appearance
        # sequential accumulation in order of appearance, as the reference's dict does
        return np.array([_seq_sum(self.probabilities[labels == f]) for f in firsts])

    def mode_genotype_support(self):
        """Genotypes congruent with the posterior mode support (reference classes.py:87-128)."""
        labels = self._support_labels()
        firsts = np.unique(labels)
        sums = self._support_sums()
        mode = firsts[np.argmax(sums)]
        idx = labels == mode
        return GenotypeSupportDistribution(self.genotypes[idx], self.probabilities[idx])

    def allele_frequencies(self, dosage=False):
        """Posterior frequency / occurrence of haplotype alleles (reference classes.py:130-166)."""
        n_gen, ploidy, n_base = self.genotypes.shape
        haps = self.genotypes.reshape(n_gen * ploidy, n_base)
        first, inv = _first_occurrence_unique(haps)
        uhaps = haps[first]
        inv = inv.reshape(n_gen, ploidy)
        ufreqs = np.zeros(len(uhaps), float)
        uoccur = np.zeros(len(uhaps), float)
        for g in range(n_gen):
            prob = self.probabilities[g]
            labs, dose = np.unique(inv[g], return_counts=True)
            ufreqs[labs] += prob * dose
            uoccur[labs] += prob
        if dosage is False:
            ufreqs /= ploidy
        return uhaps, ufreqs, uoccur


def _seq_sum(values):
    acc = 0.0
    first = True
    for v in values:
        acc = float(v) if first else acc + float(v)
        first = False
    return acc


@dataclass
class GenotypeSupportDistribution(object):
    """Genotypes with identical alleles differing only by dosage (reference classes.py:169-244)."""

    genotypes: np.ndarray
    probabilities: np.ndarray

    def alleles(self):
        return unique_rows(self.genotypes[0])

    def mode_genotype(self):
        idx = np.argmax(self.probabilities)
        return self.genotypes[idx], self.probabilities[idx]

    def call_genotype_support(self, threshold=0.95):
        if np.max(self.probabilities) >= threshold:
            idx = np.argmax(self.probabilities)
            return self.genotypes[idx], self.probabilities[idx]
        _, ploidy, n_pos = self.genotypes.shape
        result = np.zeros((ploidy, n_pos), dtype=self.genotypes.dtype) - 1
        selected = list()
        p = 0.0
        genotypes = list(self.genotypes)
        probabilities = list(self.probabilities)
        while p < threshold:
            if len(probabilities) == 0:
                break
            idx = np.argmax(probabilities)
            p += probabilities.pop(idx)
            selected.append(genotypes.pop(idx))
        alleles = reduce(_multiset_intercept, selected)
        for i, hap in enumerate(alleles):
            result[i] = hap
        return result, p


def _multiset_op(x, y, union):
    """Multiset intersection (min multiplicity) or union (max multiplicity) of the rows of two arrays; the
    result lists each distinct row `multiplicity` times, distinct rows in order of first appearance in x
    then y (the reference's Counter-based mset.intercept / mset.union)."""
    from collections import Counter

    rows = {}
    for r in x:
        rows.setdefault(r.tobytes(), r)
    cx = Counter(r.tobytes() for r in x)
    if union:
        for r in y:
            rows.setdefault(r.tobytes(), r)
    cy = Counter(r.tobytes() for r in y)
    counts = (cx | cy) if union else (cx & cy)
    out = []
    for k, v in counts.items():
        out.extend([rows[k]] * v)
    if not out:
        return np.zeros((0,) + x.shape[1:], dtype=x.dtype)
    return np.array(out, dtype=x.dtype)


def _multiset_intercept(x, y):
    return _multiset_op(x, y, union=False)


def _multiset_union(x, y):
    return _multiset_op(x, y, union=True)


@dataclass
class GenotypeMultiTrace(object):
    """Multi-chain MCMC haplotype assembler trace (reference classes.py:247-376).

    genotypes : int [n_chains, n_steps, ploidy, n_positions]; llks : float [n_chains, n_steps]."""

    genotypes: np.ndarray
    llks: np.ndarray

    def __post_init__(self):
        if (self.genotypes is not None) and (self.genotypes.shape[-1] != 0):
            self.genotypes = np.array(self.genotypes)
            self.llks = np.array(self.llks)
            assert np.ndim(self.genotypes) == 4
            assert np.ndim(self.llks) == 2
            assert self.genotypes.shape[0:2] == self.llks.shape
            self.genotypes = sort_haplotypes(self.genotypes)

    @classmethod
    def _from_sorted(cls, genotypes, llks):
        """Wrap a trace whose haplotypes are already in canonical order (what the HIP sampler writes)."""
        new = cls(None, None)
        new.genotypes = genotypes
        new.llks = llks
        return new

    def burn(self, n):
        new = type(self)(None, None)
        new.genotypes = self.genotypes[:, n:]
        new.llks = self.llks[:, n:]
        return new

    def posterior(self):
        """Posterior over phased genotypes: distinct states, probability descending (reference classes.py:316-325,
        np.flip(np.argsort(probs))).  The reference leaves the order of TIED probabilities to numpy's default,
        unstable argsort (it varies with the SIMD sort numpy dispatches to); here ties come out in descending order
        of first appearance, which is what a stable sort gives and what the device kernel implements."""
        n_chain, n_step, ploidy, n_base = self.genotypes.shape
        genotypes = self.genotypes.reshape(n_chain * n_step, ploidy, n_base)
        states, counts = unique_counts(genotypes)
        probs = counts / np.sum(counts)
        idx = np.flip(np.argsort(probs, kind="stable"))
        return PosteriorGenotypeDistribution(states[idx], probs[idx])

    def split(self):
        for genotypes, llks in zip(self.genotypes, self.llks):
            new = type(self)(None, None)
            new.genotypes = genotypes[None, ...]
            new.llks = llks[None, ...]
            yield new

    def replicate_incongruence(self, threshold=0.6):
        """0 / 1 / 2: none, incongruence, incongruence with > ploidy alleles (reference classes.py:341-376)."""
        out = 0
        posteriors = [trace.posterior() for trace in self.split()]
        chain_modes = [dist.mode_genotype_support() for dist in posteriors]
        alleles = [mode.alleles() for mode in chain_modes if mode.probabilities.sum() >= threshold]
        mode_count = len({array.tobytes() for array in alleles})
        if mode_count > 1:
            out = 1
            ploidy = len(alleles[0])
            allele_count = len(reduce(_multiset_union, alleles))
            if allele_count > ploidy:
                out = 2
        return out
