"""The command-line shell (mchap_amd/cli.py, vcfheader.py, io.py input helpers): flag parsing, per-sample parameter files,
BAM / sample tables and the VCF header block against the header of the reference's golden VCFs.  CPU only (no kernels)."""
import gzip
import os

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")


def _decl(lines):
    return [ln for ln in lines if ln.startswith(("##FILTER", "##INFO", "##FORMAT", "##contig", "##fileformat", "##phasing", "#CHROM"))]


@pytest.mark.parametrize("golden,report", [
    ("simple.output.mixed_depth.call-exact.vcf", ["SNVDP"]),
    ("simple.output.mixed_depth.call-exact.frequencies.vcf", ["AFP"]),
    ("simple.output.mixed_depth.call-exact.occurrence.vcf", ["AOP"]),
    ("simple.output.mixed_depth.call-exact.likelihoods.vcf", ["GL"]),
    ("simple.output.mixed_depth.call-exact.posteriors.vcf", ["GP"]),
    ("simple.output.mixed_depth.call-exact.frequencies.prior.vcf", ["AFPRIOR", "AFP"]),
    ("simple.output.deep.assemble.vcf", []),
])
def test_header_declarations_equal_the_reference_goldens(golden, report):
    from mchap_amd import vcfheader

    want = [ln.rstrip("\n") for ln in open(os.path.join(HERE, golden)) if ln.startswith("#")]
    got = vcfheader.header_lines("call-exact", ["x"], ["SAMPLE1", "SAMPLE2", "SAMPLE3"], [("CHR1", 60), ("CHR2", 60), ("CHR3", 60)],
                                 report=report)
    assert _decl(got) == _decl(want)


def test_flags_and_sample_files(tmp_path):
    from mchap_amd import cli, io

    a = cli.build_parser("call-exact").parse_args(["--haplotypes", "h.vcf", "--bam", "a.bam", "b.bam", "--ploidy", "4", "--use-dirmul-prior", "0.1",
                                                  "AFP", "--report", "GP", "INFO/AFP"])
    assert a.bam == ["a.bam", "b.bam"] and a.ploidy == ["4"] and a.use_dirmul_prior == ["0.1", "AFP"] and a.report == ["GP", "INFO/AFP"]
    a = cli.build_parser("assemble").parse_args(["--targets", "t.bed", "--variants", "v.vcf", "--reference", "r.fa", "--bam", "x", "--mcmc-steps", "500",
                                                "--mcmc-temperatures", "0.1", "1.0", "--mcmc-seed", "11"])
    assert a.mcmc_steps == [500] and a.mcmc_temperatures == [0.1, 1.0] and a.mcmc_seed == [11] and a.mcmc_burn == [1000]
    with pytest.raises(SystemExit):
        cli.build_parser("call").parse_args(["--bam", "x"])  # --haplotypes is required
    f = tmp_path / "ploidy.txt"
    f.write_text("S1\t4\nS2\t2\n")
    assert io.sample_values(str(f), ["S1", "S2"], int) == {"S1": 4, "S2": 2}
    assert io.sample_values("6", ["S1"], int) == 6 and io.sample_values(None, ["S1"], float) is None
    with pytest.raises(IOError):
        io.sample_values(str(f), ["S1", "S3"], int)
    from mchap_amd import vcfheader

    assert vcfheader.report_fields(["AOP", "FORMAT/GP"]) == (["AOP", "AOPSUM"], ["AOP", "GP"])
    with pytest.raises(ValueError):
        vcfheader.report_fields(["INFO/NOPE"])


def test_bam_tables_and_fasta(tmp_path):
    from mchap_amd import io

    b1, b2 = os.path.join(HERE, "simple.sample1.bam"), os.path.join(HERE, "simple.sample2.bam")
    refs, rg = io.bam_header(b1)
    assert [r[0] for r in refs] == ["CHR1", "CHR2", "CHR3"] and set(rg.values()) == {"SAMPLE1"}
    assert io.sample_bam_table([b1, b2]) == {"SAMPLE1": b1, "SAMPLE2": b2}
    lst = tmp_path / "bams.txt"
    lst.write_text(b1 + "\n" + b2 + "\n")
    assert io.sample_bam_table([str(lst)]) == {"SAMPLE1": b1, "SAMPLE2": b2}
    tab = tmp_path / "table.txt"
    tab.write_text("SAMPLE2\t%s\n" % b2)
    assert io.sample_bam_table([str(tab)]) == {"SAMPLE2": b2}
    tab.write_text("SAMPLE9\t%s\n" % b2)
    with pytest.raises(IOError):
        io.sample_bam_table([str(tab)])
    fa = tmp_path / "ref.fa.gz"
    with gzip.open(fa, "wt") as f:
        f.write(">CHR1 first\nacgt\nACGT\n>CHR2\nTTTT\n")
    assert io.read_fasta(str(fa)) == {"CHR1": "ACGTACGT", "CHR2": "TTTT"}
