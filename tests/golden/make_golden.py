#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by importing the reference itself.

Runs ONLY in the build container (needs /root/reference).  The reference is pure
Python + numba; numba and pysam are not installed here, so the hot-path bodies are run
under CPython through the identity-decorator stand-ins in tests/golden/_shim (they
contain no reference code).  Under that shim the reference draws from numpy's legacy
global MT19937 RandomState, which is what oracle/ ORC_RNG_NUMPY_MT19937 reproduces.

Output: small .npz / .json fixtures (inputs and expected outputs only - data, never
reference source).  Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "_shim"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

warnings.simplefilter("ignore")

from mchap.assemble import likelihood as ref_llk  # noqa: E402
from mchap.assemble import mutation as ref_mut  # noqa: E402
from mchap.assemble import structural as ref_str  # noqa: E402
from mchap.assemble import prior as ref_aprior  # noqa: E402
from mchap.assemble import mcmc as ref_mcmc  # noqa: E402
from mchap.assemble import snpcalling as ref_snp  # noqa: E402
from mchap.assemble.classes import GenotypeMultiTrace  # noqa: E402
from mchap.calling import exact as ref_exact  # noqa: E402
from mchap.calling import prior as ref_cprior  # noqa: E402
from mchap import jitutils as ref_jit  # noqa: E402
from mchap.encoding.integer import as_probabilistic  # noqa: E402
from mchap.testing import simulate_reads  # noqa: E402


class _NumpyNoHalf:
    """numpy stand-in for two reference modules: np.log of an int8 value.

    The reference takes np.log of int8 data in two places (assemble/mcmc.py:294, and calling/prior.py:141 when
    reached from snp_posterior with an int8 n_alleles).  Plain numpy resolves that to FLOAT16 -- an artefact of
    running without numba (numba has no float16 loops and resolves it to a wider float).  To keep the goldens free
    of that artefact the integer argument is widened to float64 first, the mathematically intended value
    ("Known reference quirks")."""

    def __getattr__(self, k):
        return getattr(np, k)

    @staticmethod
    def log(x, *a, **k):
        x = np.asarray(x)
        if x.dtype.kind in "iu":
            x = x.astype(np.float64)
        r = np.log(x, *a, **k)
        return r if r.ndim else r[()]


ref_mcmc.np = _NumpyNoHalf()
ref_cprior.np = _NumpyNoHalf()


def jdump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=None, separators=(",", ":"))
    print("wrote", name)


def zdump(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)
    print("wrote", name, sum(a.nbytes for a in arrays.values()), "bytes raw")


def random_reads(rng, R, M, A, n_alleles, gap_rate=0.25, counts=False):
    """Probabilistic reads with gaps, built with the reference's own encoder."""
    calls = np.empty((R, M), dtype=np.int8)
    for j in range(M):
        calls[:, j] = rng.integers(0, n_alleles[j], size=R)
    gaps = rng.random((R, M)) < gap_rate
    calls[gaps] = -1
    p = 0.9 + 0.0999 * rng.random((R, M))
    reads = as_probabilistic(calls, n_alleles, p)
    assert reads.shape == (R, M, A)
    rc = rng.integers(1, 6, size=R).astype(np.int64) if counts else None
    return calls, p, reads, rc


# --------------------------------------------------------------------------- G1
def gen_likelihood():
    rng = np.random.default_rng(101)
    out = {}
    n = 0
    for (K, M, A, R) in [(2, 3, 2, 5), (4, 8, 2, 40), (4, 6, 3, 17), (6, 10, 2, 33), (8, 20, 2, 50), (3, 1, 4, 9)]:
        for counts in (False, True):
            n_alleles = rng.integers(2, A + 1, size=M)
            n_alleles[0] = A
            calls, p, reads, rc = random_reads(rng, R, M, A, n_alleles, counts=counts)
            g = np.stack([rng.integers(0, n_alleles) for _ in range(K)]).astype(np.int8)
            llk = ref_llk.log_likelihood(reads, g, read_counts=rc)
            hidx = rng.integers(0, K, size=K).astype(np.int8)
            a, b = sorted(rng.choice(np.arange(M + 1), size=2, replace=False)) if M > 1 else (0, 1)
            interval = np.array([a, b])
            llk_s = ref_llk.log_likelihood_structural_change(reads, g, hidx, interval=interval, read_counts=rc)
            llk_s_full = ref_llk.log_likelihood_structural_change(reads, g, hidx, interval=None, read_counts=rc)
            pre = "c%d_" % n
            out[pre + "reads"] = reads
            out[pre + "genotype"] = g
            out[pre + "counts"] = rc if rc is not None else np.zeros(0, np.int64)
            out[pre + "hidx"] = hidx
            out[pre + "interval"] = interval
            out[pre + "llk"] = np.array([llk, llk_s, llk_s_full])
            n += 1
    out["n_cases"] = np.array(n)
    zdump("likelihood.npz", **out)


# --------------------------------------------------------------------------- G2
def reference_test_tables():
    """Known-answer tables held by the reference's own tests (pytest parametrize data)."""
    import importlib

    def params(modname, fn):
        mod = importlib.import_module(modname)
        marks = getattr(getattr(mod, fn), "pytestmark")
        res = []
        for m in marks:
            if m.name == "parametrize":
                names, values = m.args[0], m.args[1]
                values = [list(v.values) if hasattr(v, "marks") else v for v in values]  # unwrap pytest.param
                res.append({"names": names, "values": json.loads(json.dumps(values, default=lambda o: np.asarray(o).tolist()))})
        return res

    tables = {
        "assemble_dirmul_pmf": params("mchap.tests.test_assemble.test_prior", "test_log_dirichlet_multinomial_pmf"),
        "assemble_genotype_prior": params("mchap.tests.test_assemble.test_prior", "test_log_genotype_prior"),
        "structural_change": params("mchap.tests.test_assemble.test_structural", "test_structural_change"),
        "haplotype_segment_labels": params("mchap.tests.test_assemble.test_structural", "test_haplotype_segment_labels"),
        "recombination_step_options": params("mchap.tests.test_assemble.test_structural", "test_recombination_step_options"),
        "dosage_step_options": params("mchap.tests.test_assemble.test_structural", "test_dosage_step_options"),
        "get_haplotype_dosage": params("mchap.tests.test_jitutils", "test_get_haplotype_dosage"),
        "ln_equivalent_permutations": params("mchap.tests.test_jitutils", "test_ln_equivalent_permutations"),
    }
    jdump("reference_test_tables.json", tables)


def gen_priors():
    rng = np.random.default_rng(202)
    cases_a = []
    for K in (2, 4, 6, 8):
        for _ in range(6):
            # random dosage vector summing to K (first-occurrence convention: zeros allowed anywhere)
            g = np.sort(rng.integers(0, K, size=K))
            dosage = np.zeros(K, np.int8)
            for v in g:
                dosage[np.where(g == v)[0][0]] += 1
            for luh in (np.log(4.0), np.log(256.0), 8 * np.log(2.0), np.log(3.0) + np.log(2.0)):
                for F in (0.0, 0.1, 0.5, 0.9):
                    val = ref_aprior.log_genotype_prior(dosage, luh, inbreeding=F)
                    cases_a.append([dosage.tolist(), float(luh), F, float(val)])
    cases_c = []
    for K in (2, 4, 6):
        for H in (2, 5, 16):
            freqs = rng.dirichlet(np.ones(H))
            for _ in range(5):
                g = np.sort(rng.integers(0, H, size=K)).astype(np.int64)
                for F in (0.0, 0.1, 0.5):
                    v0 = ref_cprior.log_genotype_prior(g, H, inbreeding=F, frequencies=None)
                    v1 = ref_cprior.log_genotype_prior(g, H, inbreeding=F, frequencies=freqs)
                    cases_c.append([g.tolist(), H, F, None, float(v0)])
                    cases_c.append([g.tolist(), H, F, freqs.tolist(), float(v1)])
    jdump("priors.json", {"assemble": cases_a, "calling": cases_c})


# --------------------------------------------------------------------------- G3
def gen_structural():
    rng = np.random.default_rng(303)
    cases = []
    for K in (2, 4, 6, 8):
        for M in (3, 6, 10):
            for rep in range(6):
                # few distinct haplotypes so that duplicates / shared segments occur
                pool = rng.integers(0, 2, size=(3, M)).astype(np.int8)
                g = pool[rng.integers(0, 3, size=K)].copy()
                if rep % 2:
                    g[rng.integers(0, K), rng.integers(0, M)] ^= 1
                a, b = sorted(rng.choice(np.arange(M + 1), size=2, replace=False))
                interval = np.array([a, b])
                labels = ref_str.haplotype_segment_labels(g, interval)
                labels_none = ref_str.haplotype_segment_labels(g, None)
                ro = ref_str.recombination_step_options(labels)
                do = ref_str.dosage_step_options(labels)
                rn = ref_str.recombination_step_n_options(labels)
                dn = ref_str.dosage_step_n_options(labels)
                dosage = np.empty(K, np.int8)
                ref_jit.get_haplotype_dosage(dosage, g)
                copies = [int(ref_jit.count_haplotype_copies(g, h)) for h in range(K)]
                hidx = rng.integers(0, K, size=K).astype(np.int8)
                g2 = g.copy()
                ref_jit.structural_change(g2, hidx, interval)
                cases.append(
                    dict(
                        genotype=g.tolist(), interval=interval.tolist(), labels=labels.tolist(),
                        labels_none=labels_none.tolist(), recomb_options=ro.tolist(), dosage_options=do.tolist(),
                        recomb_n=int(rn), dosage_n=int(dn), dosage=dosage.tolist(), copies=copies,
                        hidx=hidx.tolist(), changed=g2.tolist(),
                        recomb_return_n=[int(ref_str.recombination_step_n_options(o)) for o in ro],
                        dosage_return_n=[int(ref_str.dosage_step_n_options(o)) for o in do],
                    )
                )
    # VCF genotype indexing
    idx_cases = []
    for K in (1, 2, 3, 4, 6, 8):
        g = np.zeros(K, np.int64)
        seq = []
        for i in range(60):
            seq.append(g.tolist())
            assert ref_jit.genotype_alleles_as_index(g) == i
            back = ref_jit.index_as_genotype_alleles(i, K)
            assert np.array_equal(back, g)
            ref_jit.increment_genotype(g)
        idx_cases.append({"ploidy": K, "genotypes": seq})
    cwr = [[n, k, int(ref_jit.comb_with_replacement(n, k))] for n in range(0, 20) for k in range(0, 10)]
    jdump("structural.json", {"cases": cases, "vcf_order": idx_cases, "comb_with_replacement": cwr})


# --------------------------------------------------------------------------- G4
def gen_presampling():
    rng = np.random.default_rng(404)
    out = {}
    n = 0
    for (K, M, A, R, F) in [(2, 4, 2, 12, None), (4, 6, 2, 40, None), (4, 5, 3, 25, 0.1), (6, 3, 4, 30, 0.0), (4, 6, 2, 0, None), (4, 4, 2, 300, 0.3)]:
        for counts in (False, True):
            n_alleles = rng.integers(2, A + 1, size=M).astype(np.int8)
            n_alleles[0] = A
            # make some positions strongly homozygous
            if R > 0:
                calls, p, reads, rc = random_reads(rng, R, M, A, n_alleles, gap_rate=0.2, counts=counts)
                calls[:, 1] = np.where(calls[:, 1] >= 0, 0, -1)
                if R >= 100:
                    calls[:, 2] = np.where(calls[:, 2] >= 0, 1, -1)
                reads = as_probabilistic(calls, n_alleles, p)
            else:
                reads = np.empty((0, M, A), float)
                rc = None
            hom = ref_mcmc._homozygosity_probabilities(reads, n_alleles, K, inbreeding=F, read_counts=rc)
            pre = "c%d_" % n
            out[pre + "reads"] = reads
            out[pre + "n_alleles"] = n_alleles
            out[pre + "counts"] = rc if rc is not None else np.zeros(0, np.int64)
            out[pre + "meta"] = np.array([K, -1.0 if F is None else F])
            out[pre + "hom"] = hom
            if R > 0:
                out[pre + "mean_dist"] = ref_mcmc._read_mean_dist(reads)
                # snp posterior at position 0
                _, probs = ref_snp.snp_posterior(reads[:, 0, :], int(n_alleles[0]), K, F, read_counts=rc)
                out[pre + "snp0"] = probs
            n += 1
    out["n_cases"] = np.array(n)
    # break distributions (scipy on the reference side)
    for (a, b) in [(1.0, 3.0), (1.0, 1.0), (2.0, 5.0)]:
        tab = np.zeros((33, 32))
        for m in range(1, 33):
            tab[m, :m] = ref_mcmc._point_beta_probabilities(m, a, b)
        out["beta_%g_%g" % (a, b)] = tab
    zdump("presampling.npz", **out)


# --------------------------------------------------------------------------- G5
def gen_exact():
    rng = np.random.default_rng(505)
    out = {}
    n = 0
    for (K, H, M, R) in [(2, 4, 4, 10), (4, 4, 5, 16), (4, 7, 6, 24), (6, 5, 5, 20), (3, 6, 4, 8)]:
        for variant in range(3):
            haps = np.unique(rng.integers(0, 2, size=(H * 4, M)).astype(np.int8), axis=0)
            rng.shuffle(haps)
            haps = haps[:H]
            H_ = len(haps)
            truth = haps[rng.integers(0, H_, size=K)]
            np.random.seed(1000 + n)
            reads = simulate_reads(truth, n_alleles=np.full(M, 2), n_reads=R, errors=True)
            rc = rng.integers(1, 5, size=R).astype(np.int64) if variant == 1 else None
            if variant == 0:
                prior = None
            elif variant == 1:
                prior = (0.2, None)
            else:
                prior = (0.1, rng.dirichlet(np.ones(H_)))
            llks = ref_exact.genotype_likelihoods(reads, K, haps, read_counts=rc)
            post = ref_exact.genotype_posteriors(llks, K, H_, prior=prior)
            post64 = ref_exact.genotype_posteriors(llks.astype(np.float64), K, H_, prior=prior)
            fr, cn, oc = ref_exact.posterior_allele_frequencies(post, K, H_)
            mode = ref_exact.posterior_mode(
                reads, K, haps, read_counts=rc, prior=prior, return_support_prob=True,
                return_posterior_frequencies=True, return_posterior_occurrence=True,
            )
            alt_g, alt_p = ref_exact.alternate_dosage_posteriors(mode[0], post)
            pre = "c%d_" % n
            out[pre + "reads"] = reads
            out[pre + "haps"] = haps
            out[pre + "counts"] = rc if rc is not None else np.zeros(0, np.int64)
            out[pre + "meta"] = np.array([K, -1.0 if prior is None else prior[0]])
            out[pre + "freqs_prior"] = prior[1] if (prior is not None and prior[1] is not None) else np.zeros(0)
            out[pre + "llks_f32"] = llks
            out[pre + "post_from_f32"] = post
            out[pre + "post_from_f64"] = post64
            out[pre + "afreq"] = np.stack([fr, cn, oc])
            out[pre + "mode_alleles"] = np.asarray(mode[0])
            out[pre + "mode_stats"] = np.array([mode[1], mode[2], mode[3]])
            out[pre + "mode_freqs"] = np.stack([mode[4], mode[5]])
            out[pre + "alt_genotypes"] = alt_g
            out[pre + "alt_probs"] = alt_p
            n += 1
    out["n_cases"] = np.array(n)
    zdump("exact.npz", **out)


# --------------------------------------------------------------------------- G6
class _Recorder:
    def __init__(self):
        self.last = None

    def __call__(self, probabilities):
        self.last = np.array(probabilities, dtype=float)
        return len(probabilities) - 1  # deterministic, irrelevant


def gen_transitions():
    rng = np.random.default_rng(606)
    out = {}
    rec = _Recorder()
    saved_m, saved_s = ref_mut.random_choice, ref_str.random_choice
    ref_mut.random_choice = rec
    ref_str.random_choice = rec
    n = 0
    try:
        for (K, M, A, R) in [(2, 4, 2, 10), (4, 6, 2, 30), (4, 5, 3, 20), (6, 4, 2, 25)]:
            for F in (None, 0.0, 0.25):
                for temp in (1.0, 0.3):
                    for counts in (False, True):
                        n_alleles = np.full(M, A, dtype=np.int8)
                        if A == 3:
                            n_alleles[1] = 2
                        calls, p, reads, rc = random_reads(rng, R, M, A, n_alleles, gap_rate=0.2, counts=counts)
                        pool = np.stack([rng.integers(0, n_alleles) for _ in range(3)]).astype(np.int8)
                        g = pool[rng.integers(0, 3, size=K)].copy()
                        llk = ref_llk.log_likelihood(reads, g, read_counts=rc)
                        luh = float(np.log(n_alleles).sum())
                        h, j = int(rng.integers(0, K)), int(rng.integers(0, M))
                        g1 = g.copy()
                        ref_mut.base_step(g1, reads, llk, h, j, int(n_alleles[j]), luh, inbreeding=F, temp=temp, read_counts=rc, cache=None)
                        pb = rec.last.copy()
                        a, b = sorted(rng.choice(np.arange(M + 1), size=2, replace=False))
                        iv = np.array([a, b])
                        res = {}
                        for st in (0, 1):
                            rec.last = None
                            g2 = g.copy()
                            ref_str.interval_step(g2, reads, llk, luh, inbreeding=F, interval=iv, step_type=st, temp=temp, read_counts=rc, cache=None)
                            res[st] = rec.last.copy() if rec.last is not None else np.zeros(0)
                        pre = "c%d_" % n
                        out[pre + "reads"] = reads
                        out[pre + "genotype"] = g
                        out[pre + "n_alleles"] = n_alleles
                        out[pre + "counts"] = rc if rc is not None else np.zeros(0, np.int64)
                        out[pre + "meta"] = np.array([K, -1.0 if F is None else F, temp, h, j, a, b, llk, luh])
                        out[pre + "p_base"] = pb
                        out[pre + "p_recomb"] = res[0]
                        out[pre + "p_dosage"] = res[1]
                        n += 1
    finally:
        ref_mut.random_choice, ref_str.random_choice = saved_m, saved_s
    out["n_cases"] = np.array(n)
    zdump("transitions.npz", **out)


# --------------------------------------------------------------------------- G8
def gen_trace_posterior():
    rng = np.random.default_rng(808)
    cases = []
    for (C, S, K, M, npool) in [(2, 40, 4, 5, 3), (1, 30, 2, 4, 4), (3, 25, 4, 3, 5), (2, 60, 6, 4, 4), (2, 20, 4, 4, 2)]:
        hpool = np.unique(rng.integers(0, 2, size=(12, M)).astype(np.int8), axis=0)
        gpool = [hpool[rng.integers(0, len(hpool), size=K)] for _ in range(npool)]
        w = rng.dirichlet(np.ones(npool) * 0.7)
        raw = np.empty((C, S, K, M), np.int8)
        for c in range(C):
            for s in range(S):
                g = gpool[rng.choice(npool, p=w)]
                raw[c, s] = g[rng.permutation(K)]
        llks = rng.normal(size=(C, S))
        trace = GenotypeMultiTrace(raw, llks)
        burn = S // 4
        post = trace.burn(burn).posterior()
        sup = post.mode_genotype_support()
        mg, mp = sup.mode_genotype()
        haps, fr, oc = post.allele_frequencies()
        inc = {str(t): int(trace.burn(burn).replicate_incongruence(t)) for t in (0.99, 0.8, 0.6, 0.3)}
        cases.append(
            dict(
                raw=raw.tolist(), burn=burn, sorted=trace.genotypes.tolist(),
                post_genotypes=post.genotypes.tolist(), post_probs=post.probabilities.tolist(),
                support_genotypes=sup.genotypes.tolist(), support_probs=sup.probabilities.tolist(),
                mode_genotype=mg.tolist(), mode_prob=float(mp), support_alleles=sup.alleles().tolist(),
                af_haps=haps.tolist(), af_freqs=fr.tolist(), af_occur=oc.tolist(), incongruence=inc,
                mode=[post.mode()[0].tolist(), float(post.mode()[1])],
            )
        )
    # constructed chains that sit on chosen supports: the three incongruence codes, including the reference's
    # `len(alleles[0])` rule (classes.py:371-375: the union of the supports is compared with the size of the FIRST
    # qualifying chain's support, not with the ploidy)
    a, b, c_, d, e = [np.array(x, np.int8) for x in ([0, 0, 0, 1], [0, 1, 1, 0], [1, 0, 1, 1], [1, 1, 0, 0], [1, 1, 1, 1])]
    designs = [
        ([[a, a, b, c_], [a, a, b, b]], 0.85),            # {a,b,c} then {a,b}: union 3 == 3 -> 1
        ([[a, a, b, b], [a, a, b, c_]], 0.85),            # {a,b} then {a,b,c}: union 3 > 2 -> 2
        ([[a, a, b, b], [a, b, b, b]], 0.7),              # same support, different dosage -> 0
        ([[a, b, c_, d], [a, b, c_, e], [a, b, c_, d]], 0.8),  # 4 + 1 = 5 > 4 -> 2
        ([[a, a, a, b], [a, a, a, c_], [b, b, b, b]], 0.75),   # {a,b}, {a,c}, {b}: union 3 > 2 -> 2
        ([[a, b, c_, c_], [a, a, b, b], [a, b, b, c_]], 0.9),  # {a,b,c}, {a,b}, {a,b,c}: union 3 == 3 -> 1
    ]
    for chains, stay in designs:
        C, S, K, M = len(chains), 40, 4, 4
        raw = np.empty((C, S, K, M), np.int8)
        for ci, main in enumerate(chains):
            for s_ in range(S):
                g = np.array(main)
                if rng.random() > stay:  # an excursion: one haplotype replaced by a random one
                    g[rng.integers(K)] = rng.integers(0, 2, size=M)
                raw[ci, s_] = g[rng.permutation(K)]
        trace = GenotypeMultiTrace(raw, rng.normal(size=(C, S)))
        burn = 4
        post = trace.burn(burn).posterior()
        sup = post.mode_genotype_support()
        mg, mp = sup.mode_genotype()
        haps, fr, oc = post.allele_frequencies()
        inc = {str(t): int(trace.burn(burn).replicate_incongruence(t)) for t in (0.99, 0.8, 0.6, 0.3)}
        cases.append(
            dict(
                raw=raw.tolist(), burn=burn, sorted=trace.genotypes.tolist(),
                post_genotypes=post.genotypes.tolist(), post_probs=post.probabilities.tolist(),
                support_genotypes=sup.genotypes.tolist(), support_probs=sup.probabilities.tolist(),
                mode_genotype=mg.tolist(), mode_prob=float(mp), support_alleles=sup.alleles().tolist(),
                af_haps=haps.tolist(), af_freqs=fr.tolist(), af_occur=oc.tolist(), incongruence=inc,
                mode=[post.mode()[0].tolist(), float(post.mode()[1])],
            )
        )
    jdump("trace_posterior.json", {"cases": cases})


# --------------------------------------------------------------------------- G9
def beta_table(n_pos, a=1.0, b=3.0):
    tab = np.zeros((n_pos + 1, n_pos))
    for m in range(1, n_pos + 1):
        tab[m, :m] = ref_mcmc._point_beta_probabilities(m, a, b)
    return tab


def gen_mcmc_traces():
    out = {}
    n = 0
    specs = [
        # name, K, M, A, R, steps, chains, F, temps, counts, seed, extra
        ("tetra_flat", 4, 5, 2, 24, 60, 2, None, (1.0,), False, 11, {}),
        ("diploid_null", 2, 6, 2, 16, 60, 2, 0.0, (1.0,), False, 12, {}),
        ("tetra_inbred_pt", 4, 5, 2, 20, 40, 1, 0.1, (0.1, 0.5, 1.0), False, 13, {}),
        ("tetra_counts", 4, 5, 2, 30, 60, 2, None, (1.0,), True, 14, {}),
        ("tri_allelic", 4, 4, 3, 20, 50, 2, 0.2, (1.0,), False, 15, {}),
        ("hexa", 6, 3, 2, 18, 40, 1, None, (1.0,), False, 16, {}),
        ("fixed_hom", 4, 6, 2, 120, 40, 2, None, (1.0,), False, 17, {"hom": True}),
        ("n_intervals", 4, 6, 2, 20, 40, 1, None, (0.5, 1.0), False, 18, {"n_intervals": 3}),
        ("zero_reads", 4, 3, 2, 0, 30, 1, None, (1.0,), False, 19, {}),
        ("initial", 4, 5, 2, 20, 30, 2, None, (1.0,), False, 20, {"initial": True}),
        ("nocache", 4, 5, 2, 24, 60, 2, None, (1.0,), False, 11, {"cache": -1, "data_seed": 0}),
        ("probs", 4, 5, 2, 20, 50, 1, 0.05, (1.0,), False, 21, {"p": (0.3, 0.8, 0.6)}),
    ]
    for (name, K, M, A, R, steps, chains, F, temps, counts, seed, extra) in specs:
        rng = np.random.default_rng(900 + extra.get("data_seed", n))
        n_alleles = np.full(M, A, dtype=np.int8)
        if A == 3:
            n_alleles[-1] = 2
        haps = np.stack([rng.integers(0, n_alleles) for _ in range(3)]).astype(np.int8)
        truth = haps[rng.integers(0, 3, size=K)]
        if R > 0:
            src = truth[rng.integers(0, K, size=R)]
            err = rng.random(src.shape) < 0.03
            calls = np.where(err, (src + 1) % n_alleles[None, :], src).astype(np.int8)
            gaps = rng.random(src.shape) < 0.2
            if extra.get("hom"):
                calls[:, 1] = 0
                calls[:, 4] = 1
                gaps[:, 1] = False
            calls[gaps] = -1
            if counts:
                p = np.full(calls.shape, 0.9976)
            else:
                p = (1 - 0.0024) * (1 - 10 ** (-rng.integers(20, 41, size=calls.shape) / 10))
            reads = as_probabilistic(calls, n_alleles, p)
            rc = None
            if counts:
                # de-duplicate as application/baseclass.py:207 does
                from mchap import mset

                reads, rc = mset.unique_counts(reads)
                rc = rc.astype(np.int64)
        else:
            reads = np.empty((0, M, A), float)
            rc = None
        kw = {}
        if "n_intervals" in extra:
            kw["n_intervals"] = extra["n_intervals"]
        if "cache" in extra:
            kw["llk_cache_threshold"] = extra["cache"]
        if "p" in extra:
            kw["recombination_step_probability"], kw["partial_dosage_step_probability"], kw["dosage_step_probability"] = extra["p"]
        model = ref_mcmc.DenovoMCMC(ploidy=K, n_alleles=n_alleles.tolist(), inbreeding=F, steps=steps, chains=chains,
                                    temperatures=temps, random_seed=seed, **kw)
        initial = None
        if extra.get("initial"):
            initial = np.stack([np.stack([rng.integers(0, n_alleles) for _ in range(K)]) for _ in range(chains)]).astype(np.int8)
        # capture the raw (unsorted) trace: GenotypeMultiTrace sorts in __post_init__
        raw = {}
        orig = ref_mcmc.GenotypeMultiTrace

        def grab(g, l):
            raw["g"], raw["l"] = np.array(g), np.array(l)
            return orig(g, l)

        ref_mcmc.GenotypeMultiTrace = grab
        try:
            model.fit(reads, read_counts=rc, initial=initial)
        finally:
            ref_mcmc.GenotypeMultiTrace = orig
        pre = "%s__" % name
        out[pre + "reads"] = reads
        out[pre + "counts"] = rc if rc is not None else np.zeros(0, np.int64)
        out[pre + "n_alleles"] = n_alleles
        out[pre + "initial"] = initial if initial is not None else np.zeros(0, np.int8)
        out[pre + "temps"] = np.array(temps)
        out[pre + "meta"] = np.array([K, steps, chains, -1.0 if F is None else F, seed, extra.get("n_intervals", 0),
                                      extra.get("cache", 100)] + list(extra.get("p", (0.5, 0.5, 1.0))))
        out[pre + "genotypes"] = raw["g"].astype(np.int8)
        out[pre + "llks"] = raw["l"]
        print(name, raw["g"].shape, "final llk", raw["l"][:, -1])
        n += 1
    out["names"] = np.array([s[0] for s in specs])
    zdump("mcmc_traces.npz", **out)


# --------------------------------------------------------------------------- G10
def gen_encoding():
    rng = np.random.default_rng(1010)
    out = {}
    calls = rng.integers(-1, 3, size=(7, 5)).astype(np.int8)
    n_alleles = np.array([2, 3, 3, 2, 3])
    calls = np.minimum(calls, n_alleles[None, :] - 1).astype(np.int8)
    p = rng.random((7, 5)) * 0.2 + 0.8
    out["calls"] = calls
    out["n_alleles"] = n_alleles
    out["p"] = p
    out["probabilistic"] = as_probabilistic(calls, n_alleles, p)
    out["probabilistic_scalar_p"] = as_probabilistic(calls, n_alleles, 0.9976)
    out["survey_example"] = as_probabilistic(np.array([[0, -1, 2]]), [2, 2, 3], 0.9)
    zdump("encoding.npz", **out)


# --------------------------------------------------------------------------- G11
def gen_call_mcmc():
    """`mchap call`: calling/mcmc.py (gibbs_options, mh_options, greedy_caller), calling/prior.py:30-113 and whole seeded
    CallingMCMC.fit traces (numpy legacy MT19937 under the shim), calling/classes.py posterior summaries."""
    from mchap.calling import mcmc as ref_cm
    from mchap.calling import prior as ref_cp
    from mchap.calling.classes import CallingMCMC

    # math.lgamma(0) raises in CPython where numba returns +inf: give the reference module numba's behaviour
    import math

    def lgamma_inf(x):
        return math.inf if x == 0 else math.lgamma(x)

    ref_cp.lgamma = lgamma_inf
    rng = np.random.default_rng(1111)
    out = {}
    n = 0
    for (K, H, M, R, prior_kind) in [(4, 5, 4, 12, None), (2, 6, 5, 20, "F"), (4, 4, 3, 15, "Ff"), (6, 5, 4, 30, "Ff"),
                                     (3, 7, 5, 25, "F0"), (4, 9, 6, 40, "Ff")]:
        haps = np.unique(rng.integers(0, 2, size=(8 * H, M)).astype(np.int8), axis=0)
        rng.shuffle(haps)
        haps = haps[:H]
        H = len(haps)
        truth = haps[rng.integers(0, H, size=K)]
        reads = simulate_reads(truth, n_reads=R, errors=False, qual=(10, 30), uniform_sample=True)
        # knock out some calls -> NaN gaps
        gap = rng.random(reads.shape[:2]) < 0.15
        reads[gap] = np.nan
        counts = rng.integers(1, 4, size=R).astype(np.int64) if K != 2 else None
        fr = rng.dirichlet(np.ones(H))
        prior = {None: None, "F": (0.2, None), "Ff": (0.15, fr), "F0": (0.0, fr)}[prior_kind]
        pre = "c%d_" % n
        out[pre + "haps"] = haps
        out[pre + "reads"] = reads
        out[pre + "counts"] = np.zeros(0, np.int64) if counts is None else counts
        out[pre + "meta"] = np.array([K, -1.0 if prior is None else prior[0], 0 if (prior is None or prior[1] is None) else 1])
        out[pre + "freqs"] = fr
        # transition vectors of both step types at random states
        states, vecs = [], []
        for t in range(4):
            g = rng.integers(0, H, size=K).astype(np.int64)
            k = int(rng.integers(0, K))
            row = []
            for fn in (ref_cm.gibbs_options, ref_cm.mh_options):
                llks, lpri, probs = np.zeros(H), np.zeros(H), np.zeros(H)
                fn(g.copy(), k, haps, reads, counts, llks, lpri, probs, prior, None)
                row.append(np.stack([llks, lpri, probs]))
            states.append(np.append(g, k))
            vecs.append(np.stack(row))
        out[pre + "states"] = np.array(states)
        out[pre + "vectors"] = np.array(vecs)  # [4][2 (gibbs, mh)][3 (llk, lprior, prob)][H]
        out[pre + "greedy"] = ref_cm.greedy_caller(haps, K, reads, counts, prior).astype(np.int64)
        # seeded fits
        for st, name in ((0, "Gibbs"), (1, "Metropolis-Hastings")):
            model = CallingMCMC(ploidy=K, haplotypes=haps, prior=prior, steps=60, chains=2, random_seed=100 + n, step_type=name)
            tr = model.fit(reads, read_counts=counts)
            out[pre + "trace%d_g" % st] = tr.genotypes.astype(np.int64)
            out[pre + "trace%d_l" % st] = tr.llks
            if st == 0:
                post = tr.burn(10).posterior()
                out[pre + "post_g"] = post.genotypes.astype(np.int64)
                out[pre + "post_p"] = post.probabilities
                m = post.mode(genotype_support=True)
                out[pre + "mode_support"] = np.append(m[0].astype(float), [m[1], m[2]])
                fq = tr.burn(10).posterior_frequencies()
                out[pre + "post_freqs"] = np.stack(fq)
                out[pre + "incongruence"] = np.array([tr.burn(10).replicate_incongruence(t) for t in (0.9, 0.6, 0.3)])
                out[pre + "as_array"] = post.as_array(H)
        # a fit from a user initial genotype
        ini = rng.integers(0, H, size=K).astype(np.int64)
        ini.sort()
        model = CallingMCMC(ploidy=K, haplotypes=haps, prior=prior, steps=25, chains=1, random_seed=7 + n)
        tr = model.fit(reads, read_counts=counts, initial=ini)
        out[pre + "initial"] = ini
        out[pre + "trace_ini_g"] = tr.genotypes.astype(np.int64)
        out[pre + "trace_ini_l"] = tr.llks
        n += 1
    # allele-prior grid
    grid = []
    for K, H in [(2, 3), (4, 5), (6, 4)]:
        fr = rng.dirichlet(np.ones(H))
        for F, f in [(0.0, None), (0.0, fr), (0.1, None), (0.35, fr)]:
            for t in range(3):
                g = rng.integers(0, H, size=K).astype(np.int64)
                k = int(rng.integers(0, K))
                grid.append(dict(g=g.tolist(), k=k, H=H, F=F, freqs=None if f is None else f.tolist(),
                                 value=float(ref_cp.log_genotype_allele_prior(g, k, H, F, f)),
                                 flat=float(ref_cp.log_genotype_allele_flat_prior(g, k))))
    out["n_cases"] = np.array(n)
    zdump("call_mcmc.npz", **out)
    jdump("call_mcmc_prior.json", {"grid": grid})


if __name__ == "__main__":
    which = sys.argv[1:] or ["likelihood", "tables", "priors", "structural", "presampling", "exact", "transitions",
                             "trace_posterior", "mcmc", "encoding", "call_mcmc"]
    fns = dict(likelihood=gen_likelihood, tables=reference_test_tables, priors=gen_priors, structural=gen_structural,
               presampling=gen_presampling, exact=gen_exact, transitions=gen_transitions,
               trace_posterior=gen_trace_posterior, mcmc=gen_mcmc_traces, encoding=gen_encoding, call_mcmc=gen_call_mcmc)
    for w in which:
        fns[w]()
