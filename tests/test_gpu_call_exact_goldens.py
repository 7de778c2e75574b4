"""GPU: the reference's RNG-free `mchap call-exact` golden VCFs (produced by the real, numba-compiled reference;
tests/test_application_call_exact.py:16-216), replayed from its own BAM / VCF test files through the GPU exact caller.
Every sample column must agree character for character (GT, GQ, SQ, DP, RCOUNT, RCALLS, MEC, MECP, GPM, SPM and the
reported arrays GL / GP / AFP / ACP / AOP / SNVDP to three decimals)."""
import os

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")

SHALLOW = ["simple.sample1.bam", "simple.sample2.bam", "simple.sample3.bam"]
MIXED = ["simple.sample1.bam", "simple.sample2.deep.bam", "simple.sample3.bam"]
SCENARIOS = [
    ("simple.output.assemble.vcf", SHALLOW, dict(), "simple.output.call-exact.vcf"),
    ("simple.output.mixed_depth.assemble.vcf", MIXED, dict(report=("SNVDP",)), "simple.output.mixed_depth.call-exact.vcf"),
    ("simple.output.mixed_depth.assemble.vcf", MIXED, dict(report=("AFP",)), "simple.output.mixed_depth.call-exact.frequencies.vcf"),
    ("simple.output.mixed_depth.assemble.vcf", MIXED, dict(report=("ACP",)), "simple.output.mixed_depth.call-exact.counts.vcf"),
    ("simple.output.mixed_depth.assemble.vcf", MIXED, dict(report=("AOP", "AOPSUM")), "simple.output.mixed_depth.call-exact.occurrence.vcf"),
    ("simple.output.mixed_depth.assemble.vcf", MIXED, dict(report=("GL",), error_rate=0.0, use_phred=True),
     "simple.output.mixed_depth.call-exact.likelihoods.vcf"),
    ("simple.output.mixed_depth.assemble.vcf", MIXED, dict(report=("GP",)), "simple.output.mixed_depth.call-exact.posteriors.vcf"),
    ("mock.input.frequencies.vcf", MIXED, dict(report=("AFPRIOR", "AFP"), prior_tag="AFP", inbreeding=0.0),
     "simple.output.mixed_depth.call-exact.frequencies.prior.vcf"),
    # --filter-input-haplotypes AFP>=0.1: rare alleles dropped, a rare reference allele masked
    ("mock.input.frequencies.vcf", MIXED, dict(report=("AFPRIOR", "AFP"), prior_tag="AFP", inbreeding=0.0, allele_filter="AFP>=0.1"),
     "simple.output.mixed_depth.call-exact.frequencies.skiprare.vcf"),
    ("mock.input.frequencies.vcf", MIXED, dict(report=("AFP", "GP"), prior_tag="AFP", inbreeding=0.0, allele_filter="AFP>=0.1"),
     "simple.output.mixed_depth.call-exact.frequencies.posteriors.skiprare.vcf"),
]


@pytest.mark.parametrize("input_vcf,bam_files,kw,golden", [sc for sc in SCENARIOS if "allele_filter" not in sc[2]])
def test_call_exact_golden_vcf(input_vcf, bam_files, kw, golden):
    import replay_call_exact as rp

    _, records = rp.read_vcf(os.path.join(HERE, input_vcf))
    samples, expect = rp.read_vcf(os.path.join(HERE, golden))
    assert samples == ["SAMPLE1", "SAMPLE2", "SAMPLE3"] and len(expect) == len(records)
    bams = {s: rp.read_bam(os.path.join(HERE, f)) for s, f in zip(samples, bam_files)}
    for rec, exp in zip(records, expect):
        assert (rec["chrom"], rec["pos"]) == (exp["chrom"], exp["pos"])
        got = rp.call_record(rec, bams, samples, **kw)
        for s in samples:
            assert got[s] == exp["samples"][s], (golden, rec["chrom"], rec["pos"], s, exp["format"])


def _golden_lines(path):
    return [ln.rstrip("\n") for ln in open(path) if ln.strip() and not ln.startswith("#")]


@pytest.mark.parametrize("input_vcf,bam_files,kw,golden", SCENARIOS)
def test_call_exact_whole_records(input_vcf, bam_files, kw, golden):
    """mchap_amd.application.call_exact: every record line (CHROM .. FILTER, INFO, FORMAT, samples) equals the golden's."""
    from mchap_amd import application

    args = dict(report=kw.get("report", ()), base_error_rate=kw.get("error_rate", 0.0024),
                use_base_phred_scores=kw.get("use_phred", False), prior_frequencies_tag=kw.get("prior_tag"),
                inbreeding=kw.get("inbreeding"), filter_input_haplotypes=kw.get("allele_filter"))
    bams = {s: os.path.join(HERE, f) for s, f in zip(["SAMPLE1", "SAMPLE2", "SAMPLE3"], bam_files)}
    got = list(application.call_exact(os.path.join(HERE, input_vcf), bams, **args))
    assert got == _golden_lines(os.path.join(HERE, golden))
