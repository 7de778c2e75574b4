"""Pin the CPU oracle (oracle/mchap_oracle.c) against vectors captured from the reference
itself (tests/golden/make_golden.py) and against the known-answer tables the reference's own
tests hold.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import binding as orc
from tests.helpers import beta_break_table

TOL = 1e-11  # fp64, identical operation order; libm vs numpy/CPython log/exp/lgamma differ by ulps


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _json(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def _rc(a):
    return None if a.size == 0 else a


def test_rng_mt19937_matches_numpy():
    for seed in (0, 1, 42, 2**32 - 1):
        np.random.seed(seed)
        expect = np.random.random(700)
        assert np.array_equal(orc.mt_doubles(seed, 700), expect)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    assert orc.philox_block([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert orc.philox_block([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert orc.philox_block([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_likelihood(golden_dir):
    z = _npz(golden_dir, "likelihood.npz")
    for i in range(int(z["n_cases"])):
        p = "c%d_" % i
        reads, g, rc = z[p + "reads"], z[p + "genotype"], _rc(z[p + "counts"])
        llk, llk_s, llk_sf = z[p + "llk"]
        assert orc.log_likelihood(reads, g, rc) == pytest.approx(llk, rel=TOL)
        assert orc.log_likelihood_structural_change(reads, g, z[p + "hidx"], z[p + "interval"], rc) == pytest.approx(llk_s, rel=TOL)
        assert orc.log_likelihood_structural_change(reads, g, z[p + "hidx"], None, rc) == pytest.approx(llk_sf, rel=TOL)
        # structural change == apply then evaluate (reference tests/test_assemble/test_likelihood.py:176-250)
        g2 = orc.structural_change(g, z[p + "hidx"], z[p + "interval"])
        assert orc.log_likelihood(reads, g2, rc) == pytest.approx(llk_s, rel=TOL)


def test_priors(golden_dir):
    d = _json(golden_dir, "priors.json")
    for dosage, luh, F, val in d["assemble"]:
        assert orc.assemble_log_genotype_prior(dosage, luh, F) == pytest.approx(val, rel=1e-10, abs=1e-11)
    for g, H, F, freqs, val in d["calling"]:
        assert orc.calling_log_genotype_prior(g, H, F, freqs) == pytest.approx(val, rel=1e-10, abs=1e-11)


def test_reference_known_answer_tables(golden_dir):
    t = _json(golden_dir, "reference_test_tables.json")
    # tensorflow-probability values, reference tests/test_assemble/test_prior.py:11-76
    (tab,) = t["assemble_dirmul_pmf"]
    for dosage, dispersion, u, prob in tab["values"]:
        # log_dirichlet_multinomial_pmf(dosage, log(dispersion), log(u)) == log_genotype_prior with that dispersion:
        # dispersion = ((1-F)/F)/u  =>  F = 1/(1+dispersion*u)
        F = 1.0 / (1.0 + dispersion * u)
        assert orc.assemble_log_genotype_prior(dosage, np.log(u), F) == pytest.approx(np.log(prob), abs=1e-9)
    (tab,) = t["assemble_genotype_prior"]
    for dosage, u, F, prob in tab["values"]:
        assert orc.assemble_log_genotype_prior(dosage, np.log(u), F) == pytest.approx(np.log(prob), abs=1e-9)
    (tab,) = t["structural_change"]
    for genotype, hidx, interval, answer in tab["values"]:
        g = np.array(genotype, np.int8)
        iv = (0, g.shape[1]) if interval is None else interval
        assert np.array_equal(orc.structural_change(g, hidx, iv), np.array(answer))
    (tab,) = t["haplotype_segment_labels"]
    for genotype, interval, answer in tab["values"]:
        assert np.array_equal(orc.haplotype_segment_labels(np.array(genotype, np.int8), interval), np.array(answer))
    (tab,) = t["recombination_step_options"]
    for labels, answer in tab["values"]:
        got = orc.step_options(np.array(labels, np.int8), 0)
        assert np.array_equal(got, np.array(answer, np.int8).reshape(got.shape))
        assert orc.step_n_options(np.array(labels, np.int8), 0) == len(answer)
    (tab,) = t["dosage_step_options"]
    for labels, answer in tab["values"]:
        got = orc.step_options(np.array(labels, np.int8), 1)
        assert np.array_equal(got, np.array(answer, np.int8).reshape(got.shape))
        assert orc.step_n_options(np.array(labels, np.int8), 1) == len(answer)
    (tab,) = t["get_haplotype_dosage"]
    for genotype, interval, answer in tab["values"]:
        if interval is None:
            assert np.array_equal(orc.get_haplotype_dosage(np.array(genotype, np.int8)), np.array(answer))


def test_structural_tables(golden_dir):
    d = _json(golden_dir, "structural.json")
    for c in d["cases"]:
        g = np.array(c["genotype"], np.int8)
        lab = orc.haplotype_segment_labels(g, c["interval"])
        assert np.array_equal(lab, np.array(c["labels"]))
        assert np.array_equal(orc.haplotype_segment_labels(g, None), np.array(c["labels_none"]))
        ro = orc.step_options(lab, 0)
        do = orc.step_options(lab, 1)
        assert ro.tolist() == c["recomb_options"]
        assert do.tolist() == c["dosage_options"]
        assert orc.step_n_options(lab, 0) == c["recomb_n"]
        assert orc.step_n_options(lab, 1) == c["dosage_n"]
        assert [orc.step_n_options(o, 0) for o in ro] == c["recomb_return_n"]
        assert [orc.step_n_options(o, 1) for o in do] == c["dosage_return_n"]
        assert orc.get_haplotype_dosage(g).tolist() == c["dosage"]
        assert [orc.count_haplotype_copies(g, h) for h in range(len(g))] == c["copies"]
        assert orc.structural_change(g, c["hidx"], c["interval"]).tolist() == c["changed"]
    for v in d["vcf_order"]:
        K = v["ploidy"]
        g = np.zeros(K, np.int64)
        for i, expect in enumerate(v["genotypes"]):
            assert g.tolist() == expect
            assert orc.genotype_alleles_as_index(g) == i
            assert orc.index_as_genotype_alleles(i, K).tolist() == expect
            g = orc.increment_genotype(g)
    for n, k, val in d["comb_with_replacement"]:
        assert orc.comb_with_replacement(n, k) == val


def test_presampling(golden_dir):
    z = _npz(golden_dir, "presampling.npz")
    for i in range(int(z["n_cases"])):
        p = "c%d_" % i
        reads, na, rc = z[p + "reads"], z[p + "n_alleles"], _rc(z[p + "counts"])
        K, F = z[p + "meta"]
        F = None if F < 0 else float(F)
        hom = orc.homozygosity_probabilities(reads, na, int(K), F, rc)
        np.testing.assert_allclose(hom, z[p + "hom"], rtol=1e-9, atol=1e-300)
        if reads.shape[0] > 0:
            np.testing.assert_allclose(orc.read_mean_dist(reads), z[p + "mean_dist"], rtol=1e-14)
            snp = orc.snp_posterior(reads[:, 0, :], int(na[0]), int(K), F, rc)
            np.testing.assert_allclose(snp, z[p + "snp0"], rtol=1e-9, atol=1e-300)
    # the break tables the python host passes in are scipy's, i.e. the reference's
    for key, (a, b) in {"beta_1_3": (1.0, 3.0), "beta_1_1": (1.0, 1.0), "beta_2_5": (2.0, 5.0)}.items():
        assert np.array_equal(beta_break_table(32, a, b), z[key])


def test_exact_caller(golden_dir):
    z = _npz(golden_dir, "exact.npz")
    for i in range(int(z["n_cases"])):
        p = "c%d_" % i
        reads, haps, rc = z[p + "reads"], z[p + "haps"], _rc(z[p + "counts"])
        K, F = z[p + "meta"]
        K = int(K)
        fr = z[p + "freqs_prior"]
        prior = None if F < 0 else (float(F), None if fr.size == 0 else fr)
        l32, l64 = orc.genotype_likelihoods(reads, K, haps, rc)
        assert l32.dtype == np.float32
        # float32 store of an fp64 value computed in the same order: identical unless the fp64 values straddle a rounding tie
        np.testing.assert_allclose(l32, z[p + "llks_f32"], rtol=2e-7)
        assert np.mean(l32 == z[p + "llks_f32"]) > 0.99
        post32 = orc.genotype_posteriors(z[p + "llks_f32"], K, len(haps), prior)
        np.testing.assert_allclose(post32, z[p + "post_from_f32"], rtol=2e-5, atol=1e-12)
        post64 = orc.genotype_posteriors(z[p + "llks_f32"].astype(np.float64), K, len(haps), prior)
        np.testing.assert_allclose(post64, z[p + "post_from_f64"], rtol=1e-9, atol=1e-300)
        f, c, o = orc.posterior_allele_frequencies(z[p + "post_from_f32"], K, len(haps))
        np.testing.assert_allclose(np.stack([f, c, o]), z[p + "afreq"], rtol=1e-12)
        alleles, mllk, mprob, sprob, freqs, occur = orc.posterior_mode(reads, K, haps, rc, prior)
        assert alleles.tolist() == z[p + "mode_alleles"].tolist()
        np.testing.assert_allclose([mllk, mprob, sprob], z[p + "mode_stats"], rtol=1e-9)
        np.testing.assert_allclose(np.stack([freqs, occur]), z[p + "mode_freqs"], rtol=1e-9, atol=1e-300)


def test_transition_vectors(golden_dir):
    """State -> proposal*acceptance probabilities, captured just before random_choice in the reference's
    base_step / interval_step: pins the accept rule without any RNG."""
    z = _npz(golden_dir, "transitions.npz")
    n_struct = 0
    for i in range(int(z["n_cases"])):
        p = "c%d_" % i
        reads, g, na, rc = z[p + "reads"], z[p + "genotype"], z[p + "n_alleles"], _rc(z[p + "counts"])
        K, F, temp, h, j, a, b, llk, luh = z[p + "meta"]
        F = None if F < 0 else float(F)
        pb, _ = orc.base_step_probabilities(reads, g, llk, int(h), int(j), int(na[int(j)]), luh, F, temp, rc)
        np.testing.assert_allclose(pb, z[p + "p_base"], rtol=1e-9, atol=1e-15)
        for st, key in ((0, "p_recomb"), (1, "p_dosage")):
            ps, _, _ = orc.interval_step_probabilities(reads, g, llk, (int(a), int(b)), st, luh, F, temp, rc)
            exp = z[p + key]
            assert len(ps) == len(exp)
            if len(exp):
                n_struct += 1
                np.testing.assert_allclose(ps, exp, rtol=1e-9, atol=1e-15)
    assert n_struct > 40


def _run_trace(z, name, rng_kind=orc.RNG_NUMPY_MT, cache=None):
    p = name + "__"
    reads, rc, na = z[p + "reads"], _rc(z[p + "counts"]), z[p + "n_alleles"]
    ini = z[p + "initial"]
    ini = None if ini.size == 0 else ini
    K, steps, chains, F, seed, n_int, cache_thr, pr, pp, pd = z[p + "meta"]
    F = None if F < 0 else float(F)
    M = reads.shape[1]
    cfg = orc.make_cfg(int(K), int(steps), int(chains), F, tuple(z[p + "temps"]), p_recomb=pr, p_partial_dosage=pp,
                       p_dosage=pd, n_intervals=None if n_int == 0 else int(n_int),
                       llk_cache_threshold=int(cache_thr) if cache is None else cache, rng_kind=rng_kind, seed=int(seed),
                       break_table=beta_break_table(M))
    return orc.denovo_fit(cfg, reads, na, rc, ini)


def test_mcmc_traces_step_for_step(golden_dir):
    """Whole seeded DenovoMCMC.fit traces of the reference (numpy MT19937 stream under the identity-njit shim)
    are reproduced step for step: bit-exact integer genotypes, llk to 1e-10."""
    z = _npz(golden_dir, "mcmc_traces.npz")
    for name in z["names"]:
        name = str(name)
        g, l, code = _run_trace(z, name)
        assert code == 0, name
        assert np.array_equal(g, z[name + "__genotypes"]), name
        np.testing.assert_allclose(l, z[name + "__llks"], rtol=1e-10, atol=1e-12, equal_nan=True, err_msg=name)


def test_cache_is_results_neutral(golden_dir):
    # reference tests/test_application_assemble.py:356,390: same output for llk_cache_threshold -1 and 10
    z = _npz(golden_dir, "mcmc_traces.npz")
    for name in ("tetra_flat", "tetra_inbred_pt", "tri_allelic"):
        g0, l0, _ = _run_trace(z, name, cache=-1)
        g1, l1, _ = _run_trace(z, name, cache=0)
        assert np.array_equal(g0, g1)
        assert np.array_equal(l0, l1)
    assert np.array_equal(z["tetra_flat__genotypes"], z["nocache__genotypes"])


def test_oracle_replays_the_call_exact_golden_vcfs():
    """The reference's RNG-free `mchap call-exact` golden VCFs (real numba outputs,
    tests/test_application_call_exact.py:16-216) replayed from its BAM / VCF test files through the ORACLE: every
    sample column character for character.  The GPU twin is tests/test_gpu_call_exact_goldens.py."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import replay_call_exact as rp
    from test_gpu_call_exact_goldens import HERE, SCENARIOS

    backend = rp.OracleBackend()
    for input_vcf, bam_files, kw, golden in SCENARIOS:
        if "allele_filter" in kw:
            continue  # (the record-level replay has no allele filter; the application test below covers those goldens)
        _, records = rp.read_vcf(os.path.join(HERE, input_vcf))
        samples, expect = rp.read_vcf(os.path.join(HERE, golden))
        bams = {s: rp.read_bam(os.path.join(HERE, f)) for s, f in zip(samples, bam_files)}
        for rec, exp in zip(records, expect):
            got = rp.call_record(rec, bams, samples, calling=backend, **kw)
            for s in samples:
                assert got[s] == exp["samples"][s], (golden, rec["chrom"], rec["pos"], s)


def test_application_call_exact_whole_records_with_the_oracle():
    """mchap_amd.application.call_exact driven by the oracle: whole record lines of the ten golden VCFs (incl. the two
    with --filter-input-haplotypes)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import replay_call_exact as rp
    from mchap_amd import application
    from test_gpu_call_exact_goldens import HERE, SCENARIOS, _golden_lines

    backend = rp.OracleBackend()
    for input_vcf, bam_files, kw, golden in SCENARIOS:
        args = dict(report=kw.get("report", ()), base_error_rate=kw.get("error_rate", 0.0024),
                    use_base_phred_scores=kw.get("use_phred", False), prior_frequencies_tag=kw.get("prior_tag"),
                    inbreeding=kw.get("inbreeding"), calling=backend, filter_input_haplotypes=kw.get("allele_filter"))
        bams = {s: os.path.join(HERE, f) for s, f in zip(["SAMPLE1", "SAMPLE2", "SAMPLE3"], bam_files)}
        got = list(application.call_exact(os.path.join(HERE, input_vcf), bams, **args))
        want = _golden_lines(os.path.join(HERE, golden))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g == w, (golden, g, w)
