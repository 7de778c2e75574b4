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

__all__ = ["CallingMCMC", "CallSummary", "GenotypeAllelesMultiTrace", "PosteriorGenotypeAllelesDistribution"]

_STEP_TYPES = {"Gibbs": 0, "Metropolis-Hastings": 1}


class _null_context(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


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

    def fit_batch_summaries(self, reads, read_counts=None, initial=None, haplotypes=None, prior=None, stream_ids=None, burn=0,
                            incongruence_threshold=0.6, max_states=512):
        """`fit_batch` with everything `mchap call` reads off a trace taken on the device (round 5): the sampler's traces stay in
        HBM and trace_posterior_kernel / trace_incongruence_kernel summarise them -- GenotypeAllelesMultiTrace.burn(burn)
        .posterior(), PosteriorGenotypeAllelesDistribution.mode(genotype_support=True) and replicate_incongruence
        (calling/classes.py:166-263, 303-362) -- so that a unit comes back as a few hundred bytes instead of its trace.
        Arguments as fit_batch.  Returns one CallSummary per unit."""
        return self.finish_batch_summaries(self.start_batch_summaries(reads, read_counts, initial, haplotypes, prior, stream_ids, burn,
                                                                      incongruence_threshold, max_states))

    def start_batch_summaries(self, reads, read_counts=None, initial=None, haplotypes=None, prior=None, stream_ids=None, burn=0,
                              incongruence_threshold=0.6, max_states=512, stream=None):
        """The first half of fit_batch_summaries: uploads, the sampler and the summary launches enqueued on `stream` (a
        torch.cuda.Stream; default: the current one) without waiting for them -- several batches (the shapes of a block of
        records) then run side by side.  Returns the handle finish_batch_summaries takes."""
        from .device import _torch

        reads = np.ascontiguousarray(reads, dtype=np.float64)
        U, R, M, A = reads.shape
        haps = np.asarray(self.haplotypes if haplotypes is None else haplotypes, dtype=np.int8)
        if haps.ndim == 2:
            haps = np.broadcast_to(haps, (U,) + haps.shape)
        haps = np.ascontiguousarray(haps)
        H = haps.shape[1]
        K, S, Cn = int(self.ploidy), int(self.steps), int(self.chains)
        burn = int(burn)
        if self.step_type not in _STEP_TYPES:
            raise ValueError('MCMC step type must be "Gibbs" or "Metropolis-Hastings"')
        n_obs = Cn * (S - burn)
        if M == 0 or K > _lib.MAX_PLOIDY:
            # (no variants: the constant trace; a ploidy beyond the device summary's: the host classes on the traces)
            return dict(done=[CallSummary.of_trace(t.burn(burn), incongruence_threshold)
                              for t in self.fit_batch(reads, read_counts, initial, haplotypes, prior, stream_ids)])
        if R == 0:
            reads = np.full((U, 1, M, A), np.nan)
            read_counts = None
            R = 1
        torch = _torch()
        dev = torch.device("cuda", torch.cuda.current_device())
        with (torch.cuda.stream(stream) if stream is not None else _null_context()):
            L = _lib.lib()
            pr = self.prior if prior is None else prior
            has = 0 if pr is None else 1
            F = fr = None
            if pr is not None:
                F = np.array(np.broadcast_to(np.asarray(pr[0], dtype=np.float64), (U,)))
                if pr[1] is not None:
                    fr = np.array(np.broadcast_to(np.asarray(pr[1], dtype=np.float64), (U, H)))
            up = lambda a, dt: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)  # noqa: E731
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
            d_reads, d_rc, d_haps = up(reads, np.float64), up(read_counts, np.int64), up(haps, np.int8)
            d_F, d_fr, d_ini = up(F, np.float64), up(fr, np.float64), up(initial, np.int64)
            if initial is not None:
                assert tuple(d_ini.shape) == (U, K)
            sid = np.ascontiguousarray(np.arange(U, dtype=np.uint64) if stream_ids is None else stream_ids, dtype=np.uint64)
            d_sid = torch.from_numpy(sid.view(np.int64)).to(dev)
            seed = self.random_seed
            if seed is None:
                seed = int(np.random.randint(0, 2**31 - 1))
            d_g = torch.empty(U * Cn * S * K, dtype=torch.int64, device=dev)
            d_l = torch.empty(U * Cn * S, dtype=torch.float64, device=dev)
            d_st = torch.empty(U, dtype=torch.int32, device=dev)
            ws = int(L.mchap_call_mcmc_workspace_bytes_for(U, R, H, K, S, Cn))
            d_ws = torch.empty(max(ws, 16), dtype=torch.uint8, device=dev)
            t_stream = torch.cuda.current_stream()
            stream = C.c_void_p(t_stream.cuda_stream)
            _lib.check(L.mchap_call_mcmc_batch_device(U, p(d_reads), R, M, A, p(d_rc), p(d_haps), H, K, has, p(d_F), p(d_fr), p(d_ini), p(d_sid), S, Cn,
                                                      _STEP_TYPES[self.step_type], C.c_uint64(int(seed) & (2**64 - 1)), p(d_g), p(d_l), p(d_st), p(d_ws),
                                                      C.c_int64(ws), stream))
            units = np.zeros(U, dtype=_lib.UNIT_DTYPE)
            units["ploidy"] = K
            units["trace_off"] = np.arange(U, dtype=np.int64) * (Cn * S * K)
            d_units = torch.from_numpy(units.view(np.uint8).reshape(-1)).to(dev)
            ms = int(max_states)
            d_words = torch.empty(U * ms * K, dtype=torch.int64, device=dev)
            d_counts = torch.empty(U * ms, dtype=torch.int32, device=dev)
            d_n = torch.empty(U, dtype=torch.int32, device=dev)
            d_stats = torch.empty(U * 2, dtype=torch.float64, device=dev)
            d_mode = torch.empty(U, dtype=torch.int32, device=dev)
            d_mw = torch.empty(U * K, dtype=torch.int64, device=dev)
            d_mc = torch.empty(U, dtype=torch.int32, device=dev)
            d_mci = torch.empty(U, dtype=torch.int32, device=dev)
            thr = C.c_double(float(incongruence_threshold))
            _lib.check(L.mchap_trace_posterior_batch_device(U, p(d_units), S, Cn, burn, p(d_g), ms, K, p(d_words), p(d_counts), p(d_n), p(d_stats),
                                                            p(d_mode), p(d_mw), p(d_mc), stream))
            _lib.check(L.mchap_call_incongruence_batch_device(U, p(d_units), S, Cn, burn, p(d_g), K, thr, p(d_mci), stream))
            return dict(torch=torch, stream=t_stream, dev=dev, L=L, p=p, U=U, K=K, S=S, Cn=Cn, H=H, burn=burn, n_obs=n_obs, ms=ms, thr=thr,
                        incongruence_threshold=incongruence_threshold, d_units=d_units, d_g=d_g, d_l=d_l, d_st=d_st, d_words=d_words,
                        d_counts=d_counts, d_n=d_n, d_stats=d_stats, d_mode=d_mode, d_mw=d_mw, d_mc=d_mc, d_mci=d_mci,
                        keep=(d_reads, d_rc, d_haps, d_F, d_fr, d_ini, d_sid, d_ws))

    def finish_batch_summaries(self, h):
        """The second half: waits for the batch, settles the units that need the listed launch (or, beyond it, the host classes) and
        returns one CallSummary per unit."""
        if "done" in h:
            return h["done"]
        with h["torch"].cuda.stream(h["stream"]):
            return self._finish(dict(h, stream=C.c_void_p(h["stream"].cuda_stream)))

    def _finish(self, v):
        torch, L, p, dev, stream = v["torch"], v["L"], v["p"], v["dev"], v["stream"]
        U, K, S, Cn, H, burn, n_obs, ms, thr = v["U"], v["K"], v["S"], v["Cn"], v["H"], v["burn"], v["n_obs"], v["ms"], v["thr"]
        incongruence_threshold = v["incongruence_threshold"]
        d_units, d_g, d_l, d_st, d_words, d_counts = v["d_units"], v["d_g"], v["d_l"], v["d_st"], v["d_words"], v["d_counts"]
        d_n, d_stats, d_mode, d_mw, d_mc, d_mci = v["d_n"], v["d_stats"], v["d_mode"], v["d_mw"], v["d_mc"], v["d_mci"]
        status = d_st.cpu().numpy()
        if (status != 0).any():
            raise _lib.MchapLibraryError("mchap_hip: the table of remembered likelihoods of a chain filled up")
        n, mci = d_n.cpu().numpy(), d_mci.cpu().numpy()
        # chains that wander through more genotypes than the batch launches keep: again with a table of every state (as many as
        # the LDS holds), a launch over those units only
        over = np.flatnonzero((n < 0) | (n > ms) | (mci < 0)).astype(np.int32)
        over_rows = {}
        if len(over):
            cap = min(n_obs, int(L.mchap_trace_posterior_max_states(K)))
            d_list = torch.from_numpy(over).to(dev)
            o_words = torch.empty(len(over) * cap * K, dtype=torch.int64, device=dev)
            o_counts = torch.empty(len(over) * cap, dtype=torch.int32, device=dev)
            _lib.check(L.mchap_trace_posterior_listed_device(len(over), p(d_list), p(d_units), S, Cn, burn, p(d_g), cap, K, p(o_words), p(o_counts),
                                                             p(d_n), p(d_stats), p(d_mode), p(d_mw), p(d_mc), stream))
            _lib.check(L.mchap_call_incongruence_listed_device(len(over), p(d_list), p(d_units), S, Cn, burn, p(d_g), min(S - burn, cap), K, thr,
                                                               p(d_mci), stream))
            n, mci = d_n.cpu().numpy(), d_mci.cpu().numpy()
            ow = o_words.cpu().numpy().reshape(len(over), cap, K)
            oc = o_counts.cpu().numpy().reshape(len(over), cap)
            over_rows = {int(u): (ow[i], oc[i], cap) for i, u in enumerate(over)}
        words = d_words.cpu().numpy().reshape(U, ms, K)
        counts = d_counts.cpu().numpy().reshape(U, ms)
        stats = d_stats.cpu().numpy().reshape(U, 2)
        mw = d_mw.cpu().numpy().reshape(U, K)
        out = []
        for u in range(U):
            w_, c_, cap_u = over_rows.get(u, (words[u], counts[u], ms))
            if n[u] < 0 or n[u] > cap_u or mci[u] < 0:
                # (more distinct genotypes than even the LDS holds: the host classes on this unit's trace)
                g = d_g[u * Cn * S * K: (u + 1) * Cn * S * K].cpu().numpy().reshape(Cn, S, K).astype(np.int32)
                lk = d_l[u * Cn * S: (u + 1) * Cn * S].cpu().numpy().reshape(Cn, S)
                out.append(CallSummary.of_trace(GenotypeAllelesMultiTrace(g, lk, H).burn(burn), incongruence_threshold))
                continue
            k = int(n[u])
            out.append(CallSummary(genotypes=w_[:k].astype(np.int32), counts=c_[:k].astype(np.int64), n_obs=n_obs, alleles=mw[u].astype(np.int32),
                                   gprob=float(stats[u, 1]), sprob=float(stats[u, 0]), mci=int(mci[u]), n_allele=H))
        return out


@dataclass
class CallSummary(object):
    """What `mchap call` reads off a unit's trace (application/call.py:95-160): the distinct genotypes after burn-in, most
    frequent first (ties as GenotypeAllelesMultiTrace.posterior), with their counts; the mode genotype of the mode support with
    its probability and the support's; the replicate incongruence code."""

    genotypes: np.ndarray   # int [n, ploidy], alleles ascending
    counts: np.ndarray      # int [n] occurrences among the n_obs recorded steps
    n_obs: int
    alleles: np.ndarray     # the mode genotype
    gprob: float
    sprob: float
    mci: int
    n_allele: int

    @classmethod
    def of_trace(cls, trace, threshold):
        """The same from a (burnt) trace on the host, by the classes below."""
        post = trace.posterior()
        alleles, gprob, sprob = post.mode(genotype_support=True)
        n_obs = trace.genotypes.shape[0] * trace.genotypes.shape[1]
        counts = np.rint(np.asarray(post.probabilities) * n_obs).astype(np.int64)
        return cls(genotypes=np.asarray(post.genotypes), counts=counts, n_obs=n_obs, alleles=np.asarray(alleles), gprob=float(gprob),
                   sprob=float(sprob), mci=int(trace.replicate_incongruence(threshold=threshold)), n_allele=trace.n_allele)

    def relabel(self, labels):
        """Alleles renamed labels[a] (an increasing map: genotypes stay ascending)."""
        labels = np.asarray(labels)
        return type(self)(labels[self.genotypes], self.counts, self.n_obs, labels[self.alleles], self.gprob, self.sprob, self.mci, int(labels.max()) + 1)

    def posterior(self):
        return PosteriorGenotypeAllelesDistribution(self.genotypes, self.counts / np.sum(self.counts))

    def posterior_frequencies(self):
        """GenotypeAllelesMultiTrace.posterior_frequencies from the distinct genotypes: the same whole-number sums (every
        recorded step counts once), divided as there."""
        g = np.asarray(self.genotypes, dtype=np.int64)
        ploidy = g.shape[1]
        w = np.repeat(self.counts.astype(float)[:, None], ploidy, axis=1)
        counts = np.bincount(g.reshape(-1), weights=w.reshape(-1), minlength=self.n_allele)
        first = np.ones(g.shape, dtype=bool)
        for i in range(1, ploidy):
            first[:, i] = (g[:, i: i + 1] != g[:, :i]).all(axis=1)
        occur = np.bincount(g[first], weights=w[first], minlength=self.n_allele)
        counts /= self.n_obs
        occur /= self.n_obs
        return counts / ploidy, counts, occur


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
