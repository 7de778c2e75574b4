"""`mchap call`'s sampler on MI355X: CallingMCMC over the genotypes of known haplotypes.

Drop-in for the reference's mchap/calling/classes.py (CallingMCMC 14-124, GenotypeAllelesMultiTrace 127-260,
PosteriorGenotypeAllelesDistribution 286-368): same dataclass fields, `fit(reads, read_counts=None, initial=None)`, same
results classes -- the sampling itself (calling/mcmc.py:15-453) runs in the HIP kernel call_mcmc_kernel through the C ABI
(mchap_call_mcmc_batch).  `fit_batch` runs many (locus x sample) units of one shape in one launch.  No CPU fallback.
"""
from dataclasses import dataclass
from math import comb

import ctypes as C
import numpy as np

from . import _lib
from .classes import Assembler, unique_counts

__all__ = ["CallingMCMC", "GenotypeAllelesMultiTrace", "PosteriorGenotypeAllelesDistribution"]

_STEP_TYPES = {"Gibbs": 0, "Metropolis-Hastings": 1}


def _vcf_index(genotypes):
    """VCF order index of every row of ascending alleles [n, K]: sum_i C(g_i + i, i + 1)."""
    g = np.asarray(genotypes, dtype=np.int64)
    idx = np.zeros(len(g), dtype=np.int64)
    for i in range(g.shape[1]):
        # C(a + i, i + 1) for allele a at position i, from a small table
        top = int(g[:, i].max(initial=0)) + 1
        table = np.array([comb(a + i, i + 1) for a in range(top + 1)], dtype=np.int64)
        idx += table[g[:, i]]
    return idx


@dataclass
class CallingMCMC(Assembler):
    """Haplotype calling by MCMC over the alleles of a genotype, given the set of known haplotypes.

    Fields as in the reference (calling/classes.py:15-21): prior = None | (inbreeding, frequencies or None);
    step_type "Gibbs" or "Metropolis-Hastings"."""

    ploidy: int
    haplotypes: np.ndarray
    prior: tuple = None
    steps: int = 1000
    chains: int = 2
    random_seed: int = None
    step_type: str = "Gibbs"

    def fit(self, reads, read_counts=None, initial=None):
        """Fit one unit; see `fit_batch`."""
        reads = np.asarray(reads)
        return self.fit_batch(reads[None], None if read_counts is None else np.asarray(read_counts)[None],
                              None if initial is None else np.asarray(initial)[None])[0]

    def fit_batch(self, reads, read_counts=None, initial=None, haplotypes=None, prior=None, stream_ids=None):
        """reads [U, R, M, A]; read_counts [U, R] or None; initial [U, K] or None (the greedy caller's genotype);
        haplotypes [U, H, M] to give every unit its own set (default: the dataclass field for all);
        prior: per-unit override (inbreeding [U], frequencies [U, H] or None); stream_ids: the units' RNG streams
        (default: their index).  Returns a list of GenotypeAllelesMultiTrace."""
        reads = np.ascontiguousarray(reads, dtype=np.float64)
        U, R, M, A = reads.shape
        haps = np.asarray(self.haplotypes if haplotypes is None else haplotypes, dtype=np.int8)
        if haps.ndim == 2:
            haps = np.broadcast_to(haps, (U,) + haps.shape)
        haps = np.array(haps)
        H = haps.shape[1]
        K, S, Cn = int(self.ploidy), int(self.steps), int(self.chains)
        if self.step_type not in _STEP_TYPES:
            raise ValueError('MCMC step type must be "Gibbs" or "Metropolis-Hastings"')  # calling/classes.py:101
        if M == 0:
            # no variants: only the reference allele (calling/classes.py:77-83)
            assert H == 1
            return [GenotypeAllelesMultiTrace(np.zeros((Cn, S, K), dtype=np.int8), np.full((Cn, S), np.nan), H) for _ in range(U)]
        if R == 0:
            # no reads: every likelihood is log(1) = 0, i.e. what one all-gap read gives (assemble/likelihood.py:54-59)
            reads = np.full((U, 1, M, A), np.nan)
            read_counts = None
            R = 1
        pr = self.prior if prior is None else prior
        has = 0 if pr is None else 1
        F = fr = None
        if pr is not None:
            F = np.array(np.broadcast_to(np.asarray(pr[0], dtype=np.float64), (U,)))
            if pr[1] is not None:
                fr = np.array(np.broadcast_to(np.asarray(pr[1], dtype=np.float64), (U, H)))
        rc = None if read_counts is None else np.ascontiguousarray(read_counts, dtype=np.int64)
        ini = None if initial is None else np.ascontiguousarray(initial, dtype=np.int64)
        if ini is not None:
            assert ini.shape == (U, K)
        sid = np.arange(U, dtype=np.uint64) if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.uint64)
        seed = self.random_seed
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1))
        g = np.zeros((U, Cn, S, K), dtype=np.int64)
        l = np.zeros((U, Cn, S), dtype=np.float64)
        status = np.zeros(U, dtype=np.int32)
        _lib.check(_lib.lib().mchap_call_mcmc_batch(
            U, _lib.ptr(reads), R, M, A, _lib.ptr(rc), _lib.ptr(haps), H, K, has, _lib.ptr(F), _lib.ptr(fr), _lib.ptr(ini),
            _lib.ptr(sid), S, Cn, _STEP_TYPES[self.step_type], C.c_uint64(int(seed) & (2**64 - 1)), _lib.ptr(g), _lib.ptr(l),
            _lib.ptr(status)))
        if (status != 0).any():
            raise _lib.MchapLibraryError("mchap_hip: the table of remembered likelihoods of a chain filled up")
        return [GenotypeAllelesMultiTrace(g[u].astype(np.int32), l[u], H) for u in range(U)]


@dataclass
class GenotypeAllelesMultiTrace(object):
    """Multi-chain trace of genotype alleles: genotypes int [n_chains, n_steps, ploidy] (alleles ascending per step),
    llks float [n_chains, n_steps], n_allele the number of known haplotypes."""

    genotypes: np.ndarray
    llks: np.ndarray
    n_allele: int

    def relabel(self, labels):
        """The trace with allele a renamed labels[a]."""
        labels = np.asarray(labels)
        return type(self)(labels[self.genotypes], self.llks, int(labels.max()) + 1)

    def burn(self, n):
        return type(self)(self.genotypes[:, n:], self.llks[:, n:], self.n_allele)

    def posterior(self):
        """Distinct genotypes of all chains with their frequencies, most frequent first (ties: see
        mchap_amd.classes.GenotypeMultiTrace.posterior)."""
        n_chain, n_step = self.genotypes.shape[:2]
        flat = self.genotypes.reshape((n_chain * n_step,) + self.genotypes.shape[2:])
        states, counts = unique_counts(flat)
        probs = counts / np.sum(counts)
        idx = np.flip(np.argsort(probs, kind="stable"))
        return PosteriorGenotypeAllelesDistribution(states[idx], probs[idx])

    def split(self):
        for c in range(len(self.genotypes)):
            yield type(self)(self.genotypes[c: c + 1], self.llks[c: c + 1], self.n_allele)

    def replicate_incongruence(self, threshold=0.6):
        """0: the chains whose mode support reaches `threshold` agree on their mode genotype; 1: they do not; 2: they
        do not and together they name more alleles than the ploidy."""
        modes = [chain.posterior().mode(genotype_support=True) for chain in self.split()]
        kept = [np.asarray(m[0]) for m in modes if m[-1] >= threshold]
        if len({g.tobytes() for g in kept}) <= 1:
            return 0
        named = set(np.concatenate(kept).tolist())
        return 2 if len(named) > len(kept[0]) else 1

    def posterior_frequencies(self):
        """(mean allele frequency, mean allele count, occurrence frequency) of every allele over all recorded steps."""
        n_chain, n_step, ploidy = self.genotypes.shape
        flat = self.genotypes.reshape(n_chain * n_step, ploidy).astype(np.int64)
        n_obs = len(flat)
        counts = np.bincount(flat.reshape(-1), minlength=self.n_allele).astype(float)
        first = np.ones(flat.shape, dtype=bool)
        for i in range(1, ploidy):
            first[:, i] = (flat[:, i: i + 1] != flat[:, :i]).all(axis=1)
        occur = np.bincount(flat[first], minlength=self.n_allele).astype(float)
        counts /= n_obs
        occur /= n_obs
        return counts / ploidy, counts, occur


@dataclass
class PosteriorGenotypeAllelesDistribution(object):
    """Posterior over genotypes of allele indices: genotypes int [n, ploidy], probabilities float [n]."""

    genotypes: np.ndarray
    probabilities: np.ndarray

    def mode(self, genotype_support=False):
        """(genotype, probability) of the most probable genotype; with genotype_support: the most probable genotype of the
        most probable SUPPORT (set of distinct alleles), its probability, and the support's summed probability."""
        probs = np.asarray(self.probabilities, dtype=float)
        if not genotype_support:
            i = int(np.argmax(probs))
            return self.genotypes[i], self.probabilities[i]
        g = np.asarray(self.genotypes)
        # support key: the distinct alleles in order of appearance, padded with -1
        K = g.shape[1]
        keys = np.full(g.shape, -1, dtype=np.int64)
        for r in range(len(g)):
            seen = []
            for a in g[r]:
                if a not in seen:
                    seen.append(int(a))
            keys[r, : len(seen)] = seen
        _, first, inv = np.unique(keys, axis=0, return_index=True, return_inverse=True)
        labels = first[inv.reshape(-1)]            # index of the first genotype with the same support
        firsts = np.unique(labels)                 # ascending == order of first appearance
        sums = np.zeros(len(firsts))
        np.add.at(sums, np.searchsorted(firsts, labels), probs)  # accumulated in genotype order
        keep = labels == firsts[int(np.argmax(sums))]
        sub_g, sub_p = g[keep], probs[keep]
        i = int(np.argmax(sub_p))
        return sub_g[i], sub_p[i], sub_p.sum()

    def as_array(self, n_alleles):
        """Probabilities over ALL genotypes of `n_alleles` alleles in VCF order (zero where unobserved)."""
        _, ploidy = self.genotypes.shape
        out = np.zeros(comb(n_alleles + ploidy - 1, ploidy) if n_alleles > 0 else 0, dtype=np.float64)
        out[_vcf_index(self.genotypes)] = self.probabilities
        return out
