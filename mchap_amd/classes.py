"""Trace and posterior containers behind the reference's API.

Same class names, method names, arguments and results as mchap/assemble/classes.py of the reference (Assembler 16-52,
PosteriorGenotypeDistribution 55-166, GenotypeSupportDistribution 169-244, GenotypeMultiTrace 247-376), written from
their documented semantics on a different representation: every haplotype row is mapped to
a small integer id (ids in order of first appearance), and the summaries are integer-table operations on those ids
instead of dictionaries keyed on `tobytes()`.  Traces produced by the HIP sampler arrive already in canonical haplotype
order (`GenotypeMultiTrace._from_sorted`), so no per-step sort runs on the host; the batched application takes these
summaries from the device kernels (mchap_amd/device.py) and uses these classes for single units and as a checker.
"""
from dataclasses import dataclass

import numpy as np

__all__ = [
    "Assembler",
    "PosteriorGenotypeDistribution",
    "GenotypeSupportDistribution",
    "GenotypeMultiTrace",
]


# ---------------------------------------------------------------------------------------------------------
# id tables
# ---------------------------------------------------------------------------------------------------------
def _row_keys(array):
    """One fixed-width opaque key per element of the outer dimension."""
    a = np.ascontiguousarray(array)
    n = len(a)
    width = a.dtype.itemsize * int(np.prod(a.shape[1:], dtype=np.int64))
    if n == 0 or width == 0:
        return np.zeros(n, dtype="V1")
    return a.reshape(n, -1).view("V%d" % width).reshape(n)


def _first_occurrence_unique(array):
    """(index of the first occurrence of each distinct element, in order of appearance; id of every element)."""
    keys = _row_keys(array)
    _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    return first[order], rank[inverse.reshape(-1)]


def unique_rows(array):
    """Distinct elements of the outer dimension in order of first appearance (what the reference's mset.unique gives)."""
    idx, _ = _first_occurrence_unique(array)
    return array[idx]


def unique_counts(array):
    """Distinct elements in order of first appearance and their counts (what mset.unique_counts gives)."""
    idx, inv = _first_occurrence_unique(array)
    return array[idx], np.bincount(inv, minlength=len(idx))


def _haplotype_ids(genotypes):
    """genotypes [n, K, M] -> (ids [n, K], table [n_haplotypes, M]): haplotype ids in order of first appearance over
    the flattened genotype list."""
    g = np.asarray(genotypes)
    n, K = g.shape[:2]
    first, ids = _first_occurrence_unique(g.reshape((n * K,) + g.shape[2:]))
    return ids.reshape(n, K), g.reshape((n * K,) + g.shape[2:])[first]


def _row_first_and_dose(ids):
    """For id rows [n, K]: mask of the first copy of each id within its row, and the number of copies of the row's id
    at every slot."""
    same = ids[:, :, None] == ids[:, None, :]          # [n, K, K]
    dose = same.sum(axis=2)
    earlier = np.tril(np.ones(ids.shape[1:] * 2, dtype=bool), -1)[None]
    first = ~(same & earlier).any(axis=2)
    return first, dose


def _support_keys(ids):
    """The ordered list of distinct ids of every row (first copies, in row order), padded with -1: two genotypes have
    the same support key iff they list the same distinct haplotypes in the same order."""
    first, _ = _row_first_and_dose(ids)
    order = np.argsort(~first, axis=1, kind="stable")  # first copies to the front, row order kept
    keys = np.take_along_axis(ids, order, axis=1)
    keys[~np.take_along_axis(first, order, axis=1)] = -1
    return keys


def sort_haplotypes(genotypes):
    """Canonical haplotype order of every genotype [..., K, M]: lexicographic, position 0 most significant (the order
    the reference establishes per step when a trace is constructed)."""
    g = np.asarray(genotypes)
    K, M = g.shape[-2:]
    flat = g.reshape(-1, K, M)
    if M == 0 or len(flat) == 0:
        return g.copy()
    # stable sorts from the least to the most significant position == a lexicographic sort per genotype
    order = np.tile(np.arange(K), (len(flat), 1))
    rows = np.arange(len(flat))[:, None]
    for j in range(M - 1, -1, -1):
        col = flat[rows, order, j]
        o = np.argsort(col, axis=1, kind="stable")
        order = np.take_along_axis(order, o, axis=1)
    return flat[rows, order].reshape(g.shape)


def _first_max(values):
    """Index of the first maximum."""
    return int(np.argmax(np.asarray(values)))


@dataclass
class Assembler(object):
    """Abstract base class for haplotype assemblers."""

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
        """(genotype, probability) of the most probable genotype."""
        i = _first_max(self.probabilities)
        return self.genotypes[i], self.probabilities[i]

    def _support_labels(self):
        """label[i] = index of the first genotype that lists the same distinct haplotypes."""
        ids, _ = _haplotype_ids(self.genotypes)
        first, inv = _first_occurrence_unique(_support_keys(ids))
        return first[inv]

    def _support_sums(self):
        """Summed probability of every support, supports in order of first appearance; each sum accumulated in
        genotype order."""
        labels = self._support_labels()
        firsts, inv = np.unique(labels, return_inverse=True)  # ascending label == order of first appearance
        sums = np.zeros(len(firsts))
        np.add.at(sums, inv, np.asarray(self.probabilities, dtype=float))  # unbuffered, in index order
        return sums

    def mode_genotype_support(self):
        """The genotypes that consist of exactly the haplotypes of the most probable support (any dosage), with
        their probabilities."""
        labels = self._support_labels()
        firsts = np.unique(labels)
        keep = labels == firsts[_first_max(self._support_sums())]
        return GenotypeSupportDistribution(self.genotypes[keep], self.probabilities[keep])

    def allele_frequencies(self, dosage=False):
        """(haplotypes, posterior frequency -- or expected dosage --, posterior probability of occurrence) of every
        haplotype of the distribution, haplotypes in order of first appearance."""
        n_gen, ploidy, _ = self.genotypes.shape
        ids, table = _haplotype_ids(self.genotypes)
        first, dose = _row_first_and_dose(ids)
        probs = np.broadcast_to(np.asarray(self.probabilities, dtype=float)[:, None], ids.shape)
        weight = np.zeros(len(table))
        occur = np.zeros(len(table))
        # one term per (genotype, distinct haplotype), added genotype by genotype
        np.add.at(weight, ids[first], probs[first] * dose[first])
        np.add.at(occur, ids[first], probs[first])
        if dosage is False:
            weight /= ploidy
        return table, weight, occur


@dataclass
class GenotypeSupportDistribution(object):
    """Genotypes over one set of haplotypes that differ only by dosage."""

    genotypes: np.ndarray
    probabilities: np.ndarray

    def alleles(self):
        return unique_rows(self.genotypes[0])

    def mode_genotype(self):
        i = _first_max(self.probabilities)
        return self.genotypes[i], self.probabilities[i]

    def call_genotype_support(self, threshold=0.95):
        """The mode genotype if it reaches `threshold`; otherwise the haplotype copies shared by the most probable
        genotypes that together reach it (multiset intersection), padded with rows of -1 up to the ploidy."""
        probs = np.asarray(self.probabilities, dtype=float)
        best = _first_max(probs)
        if probs[best] >= threshold:
            return self.genotypes[best], self.probabilities[best]
        _, ploidy, n_pos = self.genotypes.shape
        # most probable first, equal probabilities in their listed order; running total in that order
        ranked = np.argsort(-probs, kind="stable")
        total = np.cumsum(probs[ranked])
        reached = np.flatnonzero(total >= threshold)
        n_sel = int(reached[0]) + 1 if len(reached) else len(ranked)
        chosen = self.genotypes[ranked[:n_sel]]
        out = np.full((ploidy, n_pos), -1, dtype=self.genotypes.dtype)
        if n_sel == 1:
            shared = chosen[0]
        else:
            ids, table = _haplotype_ids(chosen)
            copies = np.stack([np.bincount(row, minlength=len(table)) for row in ids]).min(axis=0)
            lead = ids[0][_row_first_and_dose(ids[:1])[0][0]]  # distinct haplotypes of the first genotype, in its order
            shared = np.repeat(table[lead], copies[lead], axis=0)
        out[: len(shared)] = shared
        return out, float(total[n_sel - 1])


@dataclass
class GenotypeMultiTrace(object):
    """Multi-chain MCMC haplotype assembler trace.

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
        """Wrap a trace whose haplotypes are already in canonical order (what the HIP sampler writes): no sort."""
        new = cls(None, None)
        new.genotypes = genotypes
        new.llks = llks
        return new

    def burn(self, n):
        """The trace without the first n steps of every chain."""
        return self._from_sorted(self.genotypes[:, n:], self.llks[:, n:])

    def posterior(self):
        """Posterior over phased genotypes: the distinct states of all chains, probability descending.  Tied
        probabilities come out in descending order of first appearance (the reference leaves their order to numpy's
        unstable default argsort; DESIGN.md "Known reference quirks"), on the host and in the device kernel alike."""
        n_chain, n_step, ploidy, n_base = self.genotypes.shape
        states, counts = unique_counts(self.genotypes.reshape(n_chain * n_step, ploidy, n_base))
        probs = counts / np.sum(counts)
        idx = np.flip(np.argsort(probs, kind="stable"))
        return PosteriorGenotypeDistribution(states[idx], probs[idx])

    def split(self):
        """One single-chain trace per chain."""
        for c in range(len(self.genotypes)):
            yield self._from_sorted(self.genotypes[c: c + 1], self.llks[c: c + 1])

    def replicate_incongruence(self, threshold=0.6):
        """0: the chains whose mode support reaches `threshold` agree on it; 1: they do not; 2: they do not, and
        together they hold more haplotypes than the first of them (putative copy-number variation)."""
        supports = []
        for chain in self.split():
            support = chain.posterior().mode_genotype_support()
            if support.probabilities.sum() >= threshold:
                supports.append([h.tobytes() for h in support.alleles()])
        if len({tuple(s) for s in supports}) <= 1:
            return 0
        pooled = set().union(*supports)
        return 2 if len(pooled) > len(supports[0]) else 1
