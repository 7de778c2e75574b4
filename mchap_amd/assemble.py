"""DenovoMCMC: the reference's de-novo assembler operator on MI355X HIP kernels.

Drop-in for mchap.assemble.mcmc.DenovoMCMC (reference assemble/mcmc.py:24-161): same
dataclass fields, `fit(reads, read_counts=None, initial=None) -> GenotypeMultiTrace`, same
exceptions.  `fit_batch` runs many (locus x sample) units in one launch; `fit` is the
one-unit special case.  There is no CPU fallback: the HIP library must be built and an
MI355X visible.
"""
from dataclasses import dataclass

import ctypes as C
import os

import numpy as np
from scipy import stats as _stats

from . import _lib
from .classes import Assembler, GenotypeMultiTrace

__all__ = ["DenovoMCMC", "point_beta_probabilities", "unpack_trace"]


def point_beta_probabilities(n_base, a=1.0, b=1.0):
    """Probabilities of 0..n_base-1 break points: increments of the Beta(a, b) CDF over n_base equal
    parts (reference assemble/mcmc.py:429-452; scipy on the host, exactly as the reference does)."""
    dist = _stats.beta(a, b)
    points = np.arange(1, n_base + 1) / n_base
    probs = dist.cdf(points)
    probs[1:] = probs[1:] - probs[:-1]
    return probs


def break_table(max_pos, a, b):
    tab = np.zeros((max_pos + 1, max(max_pos, 1)), dtype=np.float64)
    for m in range(1, max_pos + 1):
        tab[m, :m] = point_beta_probabilities(m, a, b)
    return tab


def allele_bits(max_allele):
    return 1 if max_allele <= 2 else (2 if max_allele <= 4 else 3)


def unpack_trace(words, fixed_alleles, max_allele, words_per_haplotype=1):
    """Packed, sorted haplotype words [..., K] + fixed-allele template [M0] -> int8 genotypes [..., K, M0]
    (re-inserts the columns fixed as homozygous, reference assemble/mcmc.py:251-265).  words_per_haplotype = 2: the traces of
    a batch with a unit wider than 64 bits, [..., K, 2] with the most significant word first (include/mchap_hip.h)."""
    fixed_alleles = np.asarray(fixed_alleles)
    het = np.flatnonzero(fixed_alleles < 0)
    bits = allele_bits(max_allele)
    mh = len(het)
    words = np.asarray(words)
    lead = words.shape if words_per_haplotype == 1 else words.shape[:-1]
    out = np.empty(lead + (len(fixed_alleles),), dtype=np.int8)
    out[...] = np.where(fixed_alleles < 0, 0, fixed_alleles).astype(np.int8)
    if mh:
        # every sampled position's field in one pass: position jj of the mh sampled ones sits bits * (mh - 1 - jj) bits up
        sh = bits * (mh - 1 - np.arange(mh))
        mask = np.uint64((1 << bits) - 1)
        if words_per_haplotype == 1:
            out[..., het] = ((words[..., None] >> sh.astype(np.uint64)) & mask).astype(np.int8)
        else:
            assert words.shape[-1] == 2
            hi, lo = words[..., 0, None], words[..., 1, None]
            low = sh < 64
            s_lo = np.where(low, sh, 0).astype(np.uint64)                  # field starts in the low word
            up = np.where(low & (sh > 0), 64 - sh, 0).astype(np.uint64)    # ... and may run into the high one
            v_low = (lo >> s_lo) | np.where(low & (sh > 0), hi << up, np.uint64(0))
            v_high = hi >> np.where(low, 0, sh - 64).astype(np.uint64)
            out[..., het] = (np.where(low, v_low, v_high) & mask).astype(np.int8)
    return out


@dataclass
class DenovoMCMC(Assembler):
    """De novo haplotype assembly by MCMC over probabilistically encoded reads.

    Fields as in the reference (assemble/mcmc.py:26-40).  `random_seed=None` draws a seed from numpy's
    global generator; `llk_cache_threshold < 0` disables the per-chain likelihood cache, any other value enables
    it (the cache is results-neutral, as in the reference)."""

    ploidy: int
    n_alleles: list
    inbreeding: float = None
    steps: int = 1000
    chains: int = 2
    alpha: float = 1.0
    beta: float = 3.0
    n_intervals: int = None
    fix_homozygous: float = 0.999
    recombination_step_probability: float = 0.5
    partial_dosage_step_probability: float = 0.5
    dosage_step_probability: float = 1.0
    temperatures: tuple = (1.0,)
    random_seed: int = None
    llk_cache_threshold: int = 100
    kernel: int = 0  # not a reference field: 0 default, 2 lanes over chains, 3 speculative, 5 phased (same results)

    # ---- configuration shared by a batch ----
    def _cfg(self, max_pos):
        cfg = _lib.DenovoCfg()
        cfg.steps = int(self.steps)
        cfg.chains = int(self.chains)
        temps = np.sort(np.asarray(self.temperatures, dtype=np.float64))
        assert len(temps) <= _lib.MAX_TEMPS
        # reference assemble/mcmc.py:224-226
        assert temps[0] >= 0.0
        assert temps[-1] == 1.0
        cfg.n_temps = len(temps)
        for i, t in enumerate(temps):
            cfg.temperatures[i] = float(t)
        cfg.n_intervals = 0 if self.n_intervals is None else int(self.n_intervals)
        cfg.fix_homozygous = float(self.fix_homozygous)
        cfg.p_recomb = float(self.recombination_step_probability)
        cfg.p_partial_dosage = float(self.partial_dosage_step_probability)
        cfg.p_dosage = float(self.dosage_step_probability)
        seed = self.random_seed
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1))
        cfg.seed = int(seed) & (2**64 - 1)
        bt = break_table(max_pos, self.alpha, self.beta)
        cfg._keep = bt
        cfg.break_table = bt.ctypes.data
        cfg.max_pos = bt.shape[1]
        # reference assemble/mcmc.py:306-312: a negative threshold disables the likelihood cache
        cfg.llk_cache = 0 if (self.llk_cache_threshold is not None and self.llk_cache_threshold < 0) else 1
        cfg.kernel = int(self.kernel) if self.kernel else int(os.environ.get("MCHAP_HIP_KERNEL", "0"))
        # measurement / test knobs (results-neutral): MCHAP_HIP_* variables, read here -- the library reads no environment
        tuning = _lib.tuning_from_env()
        if tuning is not None:
            cfg._tuning = tuning
            cfg.tuning = C.pointer(tuning)
        return cfg

    def fit(self, reads, read_counts=None, initial=None):
        """Fit one unit; see `fit_batch`."""
        return self.fit_batch([reads], [read_counts], None if initial is None else [initial])[0]

    def fit_batch(self, reads, read_counts=None, initial=None, ploidy=None, inbreeding=None, stream_ids=None):
        """Fit a batch of independent (locus x sample) units in one kernel launch.

        reads : list of float arrays [n_reads, n_positions, max_allele]
        read_counts : list of int arrays [n_reads] or None entries
        initial : list of int arrays [chains, ploidy, n_het] or None entries
        ploidy, inbreeding : optional per-unit overrides of the dataclass fields
        stream_ids : RNG stream of each unit (default: its index); results depend only on
            (random_seed, stream id), never on batch composition or sharding.
        Returns a list of GenotypeMultiTrace."""
        n_units = len(reads)
        if read_counts is None:
            read_counts = [None] * n_units
        if initial is None:
            initial = [None] * n_units
        n_alleles = np.asarray(self.n_alleles, dtype=np.int8)
        if len(n_alleles) == 0:
            # no SNVs in the locus (reference assemble/mcmc.py:189-199 with n_het_base == 0): the constant, empty
            # genotype with nan likelihoods; nothing to sample
            out = []
            for u in range(n_units):
                rd = np.asarray(reads[u])
                assert rd.ndim == 3 and rd.shape[1] == 0  # reference assemble/mcmc.py:220
                K = int(self.ploidy if ploidy is None else ploidy[u])
                g = np.zeros((self.chains, self.steps, K, 0), dtype=np.int8)
                out.append(GenotypeMultiTrace._from_sorted(g, np.full((self.chains, self.steps), np.nan)))
            return out
        units = np.zeros(n_units, dtype=_lib.UNIT_DTYPE)
        r_parts, c_parts, i_parts = [], [], []
        r_off = c_off = i_off = t_off = l_off = f_off = 0
        shapes = []
        for u in range(n_units):
            rd = np.asarray(reads[u], dtype=np.float64)
            n_reads, n_pos, max_allele = rd.shape
            K = int(self.ploidy if ploidy is None else ploidy[u])
            if n_reads == 0:
                # reference assemble/mcmc.py:132-137: mock up a nan read
                assert len(n_alleles) == n_pos
                n_reads = 1
                rd = np.full((1, n_pos, max_allele), np.nan)
                rc = None
            else:
                rc = read_counts[u]
            assert len(n_alleles) == n_pos  # reference assemble/mcmc.py:220
            rd = np.ascontiguousarray(rd)
            r_parts.append(rd.reshape(-1))
            U = units[u]
            U["reads_off"] = r_off
            r_off += rd.size
            if rc is not None:
                rc = np.ascontiguousarray(rc, dtype=np.int64)
                assert rc.shape == (n_reads,)
                c_parts.append(rc)
                U["counts_off"] = c_off
                c_off += n_reads
            else:
                U["counts_off"] = -1
            U["nalleles_off"] = 0
            if initial[u] is not None:
                ini = np.ascontiguousarray(initial[u], dtype=np.int8)
                assert ini.ndim == 3 and ini.shape[0] == self.chains and ini.shape[1] == K
                i_parts.append(ini.reshape(-1))
                U["initial_off"] = i_off
                U["initial_n_het"] = ini.shape[2]
                i_off += ini.size
            else:
                U["initial_off"] = -1
            U["trace_off"] = t_off  # (in haplotypes here; scaled by the batch's words per haplotype below)
            t_off += self.chains * self.steps * K
            U["llk_off"] = l_off
            l_off += self.chains * self.steps
            U["fixed_off"] = f_off
            f_off += n_pos
            U["n_reads"], U["n_pos"], U["max_allele"], U["ploidy"] = n_reads, n_pos, max_allele, K
            F = self.inbreeding if inbreeding is None else inbreeding[u]
            U["inbreeding"] = np.nan if F is None else float(F)
            U["stream_id"] = u if stream_ids is None else int(stream_ids[u])
            shapes.append((n_pos, max_allele, K, None if initial[u] is None else ini.shape))
        cfg = self._cfg(len(n_alleles))
        # uint64 words per haplotype of the traces: 1, or 2 for a batch with a unit of more than 62 SNVs / 64 bits per haplotype
        wph = int(_lib.lib().mchap_denovo_trace_words_per_haplotype(C.byref(cfg), n_units, _lib.ptr(units)))
        if wph < 1:
            _lib.check(_lib.lib().mchap_denovo_sampler_name(C.byref(cfg), n_units, _lib.ptr(units), C.create_string_buffer(8), 8))
            raise NotImplementedError("mchap_hip: unsupported unit shape")
        units["trace_off"] *= wph
        t_off *= wph
        reads_flat = np.concatenate(r_parts)
        counts_flat = np.concatenate(c_parts) if c_parts else None
        init_flat = np.concatenate(i_parts) if i_parts else None
        trace = np.zeros(t_off, dtype=np.uint64)
        llks = np.zeros(l_off, dtype=np.float64)
        fixed = np.zeros(f_off, dtype=np.int8)
        status = np.zeros(n_units, dtype=np.int32)
        L = _lib.lib()
        self.last_sampler = _lib.sampler_name(cfg, units)  # (not a reference field: which kernel(s) the batch ran on)
        cfg.cache_epoch = _lib.next_cache_epoch()
        rc = L.mchap_denovo_fit_batch(
            C.byref(cfg), n_units, _lib.ptr(units), _lib.ptr(reads_flat), C.c_int64(reads_flat.size),
            _lib.ptr(counts_flat), C.c_int64(0 if counts_flat is None else counts_flat.size),
            _lib.ptr(n_alleles), C.c_int64(n_alleles.size), _lib.ptr(init_flat),
            C.c_int64(0 if init_flat is None else init_flat.size), _lib.ptr(trace), C.c_int64(trace.size),
            _lib.ptr(llks), C.c_int64(llks.size), _lib.ptr(fixed), C.c_int64(fixed.size), _lib.ptr(status))
        _lib.check(rc)
        out = []
        for u in range(n_units):
            st = int(status[u])
            if st == _lib.UNIT_NAN_LLK:
                raise ValueError("Encountered log likelihood of nan")  # reference assemble/mcmc.py:330-331
            if st == _lib.UNIT_BREAKS:
                raise ValueError("breaks must be smaller then n")  # reference assemble/structural.py:49-50
            if st < 0:
                raise NotImplementedError("mchap_hip: unit %d exceeds the packed haplotype width" % u)
            # reference assemble/mcmc.py:207; checked on the device BEFORE `initial` is read
            assert st != _lib.UNIT_BAD_INITIAL, "initial.shape != (ploidy, n_het_base)"
            n_pos, max_allele, K, ini_shape = shapes[u]
            U = units[u]
            fx = fixed[U["fixed_off"]: U["fixed_off"] + n_pos]
            w = trace[U["trace_off"]: U["trace_off"] + self.chains * self.steps * K * wph]
            w = w.reshape((self.chains, self.steps, K) + ((wph,) if wph > 1 else ()))
            g = unpack_trace(w, fx, max_allele, wph)
            lk = llks[U["llk_off"]: U["llk_off"] + self.chains * self.steps].reshape(self.chains, self.steps).copy()
            out.append(GenotypeMultiTrace._from_sorted(g, lk))
        return out
