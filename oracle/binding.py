"""ctypes binding of the CPU ORACLE (oracle/mchap_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package (mchap_amd).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libmchap_oracle.so")

MAX_TEMPS = 16
RNG_PHILOX = 0
RNG_NUMPY_MT = 1

f64p = C.POINTER(C.c_double)
f32p = C.POINTER(C.c_float)
i8p = C.POINTER(C.c_int8)
i64p = C.POINTER(C.c_int64)


class DenovoCfg(C.Structure):
    _fields_ = [
        ("ploidy", C.c_int32),
        ("steps", C.c_int32),
        ("chains", C.c_int32),
        ("n_temps", C.c_int32),
        ("temperatures", C.c_double * MAX_TEMPS),
        ("inbreeding", C.c_double),
        ("fix_homozygous", C.c_double),
        ("p_recomb", C.c_double),
        ("p_partial_dosage", C.c_double),
        ("p_dosage", C.c_double),
        ("n_intervals", C.c_int32),
        ("llk_cache_threshold", C.c_int32),
        ("rng_kind", C.c_int32),
        ("reserved", C.c_int32),
        ("seed", C.c_uint64),
        ("stream_id", C.c_uint64),
        ("break_table", f64p),
    ]


class Stats(C.Structure):
    _fields_ = [("llk_evals", C.c_int64), ("llk_cache_hits", C.c_int64), ("mutation_evals", C.c_int64),
                ("structural_evals", C.c_int64)]


def build(force=False):
    src = os.path.join(HERE, "mchap_oracle.c")
    hdr = os.path.join(HERE, "mchap_oracle.h")
    if (not force and os.path.exists(SO) and os.path.getmtime(SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return SO
    subprocess.check_call(["make", "-C", HERE, "-B"], stdout=subprocess.DEVNULL)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.orc_log_likelihood.restype = C.c_double
        _lib.orc_log_likelihood_structural_change.restype = C.c_double
        _lib.orc_assemble_log_genotype_prior.restype = C.c_double
        _lib.orc_calling_log_genotype_prior.restype = C.c_double
        _lib.orc_comb_with_replacement.restype = C.c_int64
        _lib.orc_genotype_alleles_as_index.restype = C.c_int64
        _lib.orc_add_log_prob.restype = C.c_double
        _lib.orc_philox_double.restype = C.c_double
        _lib.orc_philox_interval.restype = C.c_uint32
        _lib.orc_version.restype = C.c_char_p
        _lib.orc_log_genotype_allele_prior.restype = C.c_double
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def _i64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def _nanF(F):
    return float("nan") if F is None else float(F)


def log_likelihood(reads, genotype, read_counts=None):
    reads = _f64(reads)
    g = _i8(genotype)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    return lib().orc_log_likelihood(_p(reads, f64p), R, M, A, _p(g, i8p), g.shape[0], _p(rc, i64p))


def log_likelihood_structural_change(reads, genotype, hidx, interval=None, read_counts=None):
    reads = _f64(reads)
    g = _i8(genotype)
    hi = _i8(hidx)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    a, b = (0, M) if interval is None else (int(interval[0]), int(interval[1]))
    return lib().orc_log_likelihood_structural_change(_p(reads, f64p), R, M, A, _p(g, i8p), g.shape[0], _p(hi, i8p), a, b,
                                                      _p(rc, i64p))


def assemble_log_genotype_prior(dosage, log_unique_haplotypes, inbreeding):
    d = _i8(dosage)
    return lib().orc_assemble_log_genotype_prior(_p(d, i8p), len(d), C.c_double(log_unique_haplotypes), C.c_double(inbreeding))


def calling_log_genotype_prior(genotype, unique_haplotypes, inbreeding, frequencies=None):
    g = _i64(genotype)
    fr = None if frequencies is None else _f64(frequencies)
    return lib().orc_calling_log_genotype_prior(_p(g, i64p), len(g), C.c_int64(unique_haplotypes), C.c_double(inbreeding),
                                                _p(fr, f64p))


def get_haplotype_dosage(genotype):
    g = _i8(genotype)
    d = np.zeros(g.shape[0], np.int8)
    lib().orc_get_haplotype_dosage(_p(d, i8p), _p(g, i8p), g.shape[0], g.shape[1])
    return d


def count_haplotype_copies(genotype, h):
    g = _i8(genotype)
    return lib().orc_count_haplotype_copies(_p(g, i8p), g.shape[0], g.shape[1], int(h))


def structural_change(genotype, hidx, interval):
    g = _i8(genotype).copy()
    hi = _i8(hidx)
    lib().orc_structural_change(_p(g, i8p), g.shape[0], g.shape[1], _p(hi, i8p), int(interval[0]), int(interval[1]))
    return g


def haplotype_segment_labels(genotype, interval=None):
    g = _i8(genotype)
    K, M = g.shape
    a, b = (0, M) if interval is None else (int(interval[0]), int(interval[1]))
    lab = np.zeros((K, 2), np.int8)
    lib().orc_haplotype_segment_labels(_p(g, i8p), K, M, a, b, _p(lab, i8p))
    return lab


def step_options(labels, step_type):
    lab = _i8(labels)
    K = lab.shape[0]
    buf = np.zeros((K * K, K, 2), np.int8)
    fn = lib().orc_recombination_step_options if step_type == 0 else lib().orc_dosage_step_options
    n = fn(_p(lab, i8p), K, _p(buf, i8p))
    return buf[:n].copy()


def step_n_options(labels, step_type):
    lab = _i8(labels)
    fn = lib().orc_recombination_step_n_options if step_type == 0 else lib().orc_dosage_step_n_options
    return fn(_p(lab, i8p), lab.shape[0])


def increment_genotype(g):
    g = _i64(g).copy()
    lib().orc_increment_genotype(_p(g, i64p), len(g))
    return g


def comb_with_replacement(n, k):
    return lib().orc_comb_with_replacement(C.c_int64(n), C.c_int64(k))


def genotype_alleles_as_index(g):
    g = _i64(g)
    return lib().orc_genotype_alleles_as_index(_p(g, i64p), len(g))


def index_as_genotype_alleles(index, ploidy):
    out = np.zeros(ploidy, np.int64)
    lib().orc_index_as_genotype_alleles(C.c_int64(index), ploidy, _p(out, i64p))
    return out


def base_step_probabilities(reads, genotype, llk, h, j, n_alleles_j, luh, inbreeding, temp, read_counts=None):
    reads = _f64(reads)
    g = _i8(genotype)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    probs = np.zeros(n_alleles_j)
    llks = np.zeros(n_alleles_j)
    rcode = lib().orc_base_step_probabilities(_p(reads, f64p), R, M, A, _p(g, i8p), g.shape[0], C.c_double(llk), int(h), int(j),
                                              int(n_alleles_j), C.c_double(luh), C.c_double(_nanF(inbreeding)),
                                              C.c_double(temp), _p(rc, i64p), _p(probs, f64p), _p(llks, f64p))
    assert rcode == 0
    return probs, llks


def interval_step_probabilities(reads, genotype, llk, interval, step_type, luh, inbreeding, temp, read_counts=None):
    reads = _f64(reads)
    g = _i8(genotype)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    K = g.shape[0]
    probs = np.zeros(K * K + 1)
    llks = np.zeros(K * K + 1)
    opts = np.zeros((K * K, K, 2), np.int8)
    n = lib().orc_interval_step_probabilities(_p(reads, f64p), R, M, A, _p(g, i8p), K, C.c_double(llk), int(interval[0]),
                                              int(interval[1]), int(step_type), C.c_double(luh),
                                              C.c_double(_nanF(inbreeding)), C.c_double(temp), _p(rc, i64p),
                                              _p(probs, f64p), _p(llks, f64p), _p(opts, i8p))
    assert n >= 0
    if n == 0:
        return np.zeros(0), np.zeros(0), opts[:0]
    return probs[: n + 1].copy(), llks[: n + 1].copy(), opts[:n].copy()


def snp_posterior(read_probs, n_alleles, ploidy, inbreeding=None, read_counts=None):
    rp = _f64(read_probs)
    rc = _i64(read_counts)
    R, A = rp.shape
    u = comb_with_replacement(n_alleles, ploidy)
    probs = np.zeros(u)
    lib().orc_snp_posterior(_p(rp, f64p), R, A, int(n_alleles), int(ploidy), C.c_double(_nanF(inbreeding)), _p(rc, i64p),
                            _p(probs, f64p))
    return probs


def homozygosity_probabilities(reads, n_alleles, ploidy, inbreeding=None, read_counts=None):
    reads = _f64(reads)
    na = _i8(n_alleles)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    out = np.zeros((M, A))
    lib().orc_homozygosity_probabilities(_p(reads, f64p), R, M, A, _p(na, i8p), int(ploidy), C.c_double(_nanF(inbreeding)),
                                         _p(rc, i64p), _p(out, f64p))
    return out


def read_mean_dist(reads):
    reads = _f64(reads)
    R, M, A = reads.shape
    out = np.zeros((M, A))
    lib().orc_read_mean_dist(_p(reads, f64p), R, M, A, _p(out, f64p))
    return out


def make_cfg(ploidy, steps=1000, chains=2, inbreeding=None, temperatures=(1.0,), fix_homozygous=0.999,
             p_recomb=0.5, p_partial_dosage=0.5, p_dosage=1.0, n_intervals=None, llk_cache_threshold=100,
             rng_kind=RNG_PHILOX, seed=0, stream_id=0, break_table=None):
    cfg = DenovoCfg()
    cfg.ploidy, cfg.steps, cfg.chains = int(ploidy), int(steps), int(chains)
    temps = np.sort(np.asarray(temperatures, dtype=float))
    cfg.n_temps = len(temps)
    for i, t in enumerate(temps):
        cfg.temperatures[i] = float(t)
    cfg.inbreeding = _nanF(inbreeding)
    cfg.fix_homozygous = float(fix_homozygous)
    cfg.p_recomb, cfg.p_partial_dosage, cfg.p_dosage = float(p_recomb), float(p_partial_dosage), float(p_dosage)
    cfg.n_intervals = 0 if n_intervals is None else int(n_intervals)
    cfg.llk_cache_threshold = int(llk_cache_threshold)
    cfg.rng_kind = int(rng_kind)
    cfg.seed = int(seed)
    cfg.stream_id = int(stream_id)
    if break_table is not None:
        cfg._bt = _f64(break_table)  # keep alive
        cfg.break_table = _p(cfg._bt, f64p)
    return cfg


def denovo_fit(cfg, reads, n_alleles, read_counts=None, initial=None, want_stats=False):
    """Returns (genotypes int8 [C,S,K,M], llks f64 [C,S], rc[, stats])."""
    reads = _f64(reads)
    R, M, A = reads.shape
    na = _i8(n_alleles)
    rc = _i64(read_counts)
    ini = None if initial is None else _i8(initial)
    g = np.zeros((cfg.chains, cfg.steps, cfg.ploidy, M), np.int8)
    l = np.zeros((cfg.chains, cfg.steps))
    st = Stats()
    code = lib().orc_denovo_fit(C.byref(cfg), _p(reads, f64p), R, M, A, _p(rc, i64p), _p(na, i8p), _p(ini, i8p),
                                _p(g, i8p), _p(l, f64p), C.byref(st))
    if want_stats:
        return g, l, code, st
    return g, l, code


def denovo_fit_batch(cfg, reads, n_alleles, read_counts=None, n_threads=0, keep_traces=True):
    """reads [U,R,M,A]. Returns (genotypes [U,C,S,K,M] or None, llks [U,C,S], rc, stats)."""
    reads = _f64(reads)
    U, R, M, A = reads.shape
    na = _i8(n_alleles)
    rc = _i64(read_counts)
    g = l = None
    if keep_traces:
        g = np.zeros((U, cfg.chains, cfg.steps, cfg.ploidy, M), np.int8)
        l = np.zeros((U, cfg.chains, cfg.steps))
        g.fill(0)
        l.fill(0.0)  # touch the pages outside the timed parallel region
    st = Stats()
    code = lib().orc_denovo_fit_batch(C.byref(cfg), U, int(n_threads), _p(reads, f64p), R, M, A, _p(rc, i64p), _p(na, i8p),
                                      _p(g, i8p), _p(l, f64p), C.byref(st))
    return g, l, code, st


def genotype_likelihoods(reads, ploidy, haplotypes, read_counts=None):
    reads = _f64(reads)
    haps = _i8(haplotypes)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    G = comb_with_replacement(len(haps), ploidy)
    out = np.zeros(G, np.float32)
    out64 = np.zeros(G)
    code = lib().orc_genotype_likelihoods(_p(reads, f64p), R, M, A, int(ploidy), _p(haps, i8p), len(haps), _p(rc, i64p),
                                          _p(out, f32p), _p(out64, f64p))
    assert code == 0
    return out, out64


def genotype_posteriors(llks, ploidy, n_alleles, prior=None):
    has = 0 if prior is None else 1
    F = 0.0 if prior is None else float(prior[0])
    fr = None if (prior is None or prior[1] is None) else _f64(prior[1])
    G = len(llks)
    out = np.zeros(G)
    if llks.dtype == np.float32:
        l = np.ascontiguousarray(llks)
        lib().orc_genotype_posteriors_f32(_p(l, f32p), C.c_int64(G), int(ploidy), int(n_alleles), has, C.c_double(F),
                                          _p(fr, f64p), _p(out, f64p))
    else:
        l = _f64(llks)
        lib().orc_genotype_posteriors_f64(_p(l, f64p), C.c_int64(G), int(ploidy), int(n_alleles), has, C.c_double(F),
                                          _p(fr, f64p), _p(out, f64p))
    return out


def posterior_allele_frequencies(posteriors, ploidy, n_alleles):
    p = _f64(posteriors)
    fr, cn, oc = np.zeros(n_alleles), np.zeros(n_alleles), np.zeros(n_alleles)
    lib().orc_posterior_allele_frequencies_f64(_p(p, f64p), C.c_int64(len(p)), int(ploidy), int(n_alleles), _p(fr, f64p),
                                               _p(cn, f64p), _p(oc, f64p))
    return fr, cn, oc


def posterior_mode(reads, ploidy, haplotypes, read_counts=None, prior=None):
    reads = _f64(reads)
    haps = _i8(haplotypes)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    H = len(haps)
    has = 0 if prior is None else 1
    F = 0.0 if prior is None else float(prior[0])
    fr = None if (prior is None or prior[1] is None) else _f64(prior[1])
    alleles = np.zeros(ploidy, np.int64)
    mllk, mprob, sprob = C.c_double(), C.c_double(), C.c_double()
    freqs, occur = np.zeros(H), np.zeros(H)
    code = lib().orc_posterior_mode(_p(reads, f64p), R, M, A, int(ploidy), _p(haps, i8p), H, _p(rc, i64p), has, C.c_double(F),
                                    _p(fr, f64p), _p(alleles, i64p), C.byref(mllk), C.byref(mprob), C.byref(sprob),
                                    _p(freqs, f64p), _p(occur, f64p))
    assert code == 0
    return alleles, mllk.value, mprob.value, sprob.value, freqs, occur


def philox_double(seed, stream_id, substream, n):
    return lib().orc_philox_double(C.c_uint64(seed), C.c_uint64(stream_id), C.c_uint32(substream), C.c_uint64(n))


def philox_interval(seed, stream_id, substream, n, mx):
    return lib().orc_philox_interval(C.c_uint64(seed), C.c_uint64(stream_id), C.c_uint32(substream), C.c_uint64(n),
                                     C.c_uint32(mx))


def philox_block(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def mt_doubles(seed, n):
    out = np.zeros(n)
    lib().orc_mt_doubles(C.c_uint32(seed), n, _p(out, f64p))
    return out


# ---- `mchap call` sampler over known haplotypes (calling/mcmc.py) ----
def _prior_args(prior):
    has = 0 if prior is None else 1
    F = 0.0 if prior is None else float(prior[0])
    fr = None if (prior is None or prior[1] is None) else _f64(prior[1])
    return has, F, fr


def log_genotype_allele_prior(genotype, variable_allele, unique_haplotypes, prior=None):
    """calling/prior.py:30-113; prior None -> log_genotype_allele_flat_prior."""
    g = _i64(genotype)
    has, F, fr = _prior_args(prior)
    return lib().orc_log_genotype_allele_prior(_p(g, i64p), len(g), int(variable_allele), C.c_int64(unique_haplotypes), has,
                                               C.c_double(F), _p(fr, f64p))


def call_step_options(reads, haplotypes, genotype, variable_allele, step_type, read_counts=None, prior=None):
    """(llks, lpriors, probabilities) of gibbs_options (step_type 0) / mh_options (1), calling/mcmc.py:15-229."""
    reads = _f64(reads)
    haps = _i8(haplotypes)
    rc = _i64(read_counts)
    g = _i64(genotype)
    R, M, A = reads.shape
    H = len(haps)
    has, F, fr = _prior_args(prior)
    llks, lpriors, probs = np.zeros(H), np.zeros(H), np.zeros(H)
    code = lib().orc_call_step_options(_p(reads, f64p), R, M, A, _p(rc, i64p), _p(haps, i8p), H, _p(g, i64p), len(g),
                                       int(variable_allele), int(step_type), has, C.c_double(F), _p(fr, f64p),
                                       _p(llks, f64p), _p(lpriors, f64p), _p(probs, f64p))
    assert code == 0
    return llks, lpriors, probs


def greedy_caller(reads, haplotypes, ploidy, read_counts=None, prior=None):
    reads = _f64(reads)
    haps = _i8(haplotypes)
    rc = _i64(read_counts)
    R, M, A = reads.shape
    has, F, fr = _prior_args(prior)
    out = np.zeros(ploidy, np.int64)
    code = lib().orc_greedy_caller(_p(reads, f64p), R, M, A, _p(rc, i64p), _p(haps, i8p), len(haps), int(ploidy), has,
                                   C.c_double(F), _p(fr, f64p), _p(out, i64p))
    assert code == 0
    return out


def call_mcmc(reads, haplotypes, ploidy, steps=1000, chains=2, step_type=0, read_counts=None, prior=None, initial=None,
              rng_kind=RNG_PHILOX, seed=0, stream_id=0):
    """CallingMCMC.fit (calling/classes.py:62-124): (genotypes int64 [chains, steps, ploidy], llks [chains, steps])."""
    reads = _f64(reads)
    haps = _i8(haplotypes)
    rc = _i64(read_counts)
    ini = _i64(initial)
    R, M, A = reads.shape
    has, F, fr = _prior_args(prior)
    g = np.zeros((chains, steps, ploidy), np.int64)
    l = np.zeros((chains, steps))
    code = lib().orc_call_mcmc(_p(reads, f64p), R, M, A, _p(rc, i64p), _p(haps, i8p), len(haps), int(ploidy), has,
                               C.c_double(F), _p(fr, f64p), int(steps), int(chains), int(step_type), _p(ini, i64p),
                               int(rng_kind), C.c_uint64(seed), C.c_uint64(stream_id), _p(g, i64p), _p(l, f64p))
    assert code == 0, code
    return g, l
