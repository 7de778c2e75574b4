"""The command-line shell (mchap_amd/cli.py, vcfheader.py, io.py input helpers): flag parsing, per-sample parameter files,
BAM / sample tables and the VCF header block against the header of the reference's golden VCFs.  CPU only (no kernels)."""
import gzip
import os

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")


def _decl(lines):
    # (the FILTER=LIMIT declaration is this build's own: records beyond its shape limits, which the reference does not have)
    return [ln for ln in lines if ln.startswith(("##FILTER", "##INFO", "##FORMAT", "##contig", "##fileformat", "##phasing", "#CHROM"))
            and not ln.startswith("##FILTER=<ID=LIMIT,")]


@pytest.mark.parametrize("golden,report", [
    ("simple.output.mixed_depth.call-exact.vcf", ["SNVDP"]),
    ("simple.output.mixed_depth.call-exact.frequencies.vcf", ["AFP"]),
    ("simple.output.mixed_depth.call-exact.occurrence.vcf", ["AOP", "AOPSUM"]),
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
    assert a.mcmc_steps == [500] and a.mcmc_temperatures == ["0.1", "1.0"] and a.mcmc_seed == [11] and a.mcmc_burn == [1000]
    assert cli.build_parser("call").parse_args(["--bam", "x"]).mcmc_seed == [42]  # the reference's default seed
    with pytest.raises(ValueError, match="--haplotypes"):
        cli.run(["mchap_amd", "call", "--bam", os.path.join(HERE, "simple.sample1.bam")])
    f = tmp_path / "ploidy.txt"
    f.write_text("S1\t4\nS2\t2\n")
    assert io.sample_values(str(f), ["S1", "S2"], int) == {"S1": 4, "S2": 2}
    assert io.sample_values("6", ["S1"], int) == 6 and io.sample_values(None, ["S1"], float) is None
    with pytest.raises(IOError):
        io.sample_values(str(f), ["S1", "S3"], int)
    from mchap_amd import vcfheader

    assert vcfheader.report_fields(["AOP", "FORMAT/GP"]) == (["AOP"], ["AOP", "GP"])
    # whatever the order of the arguments: the order of the reference's field tables (infofields.py:125, formatfields.py:157)
    assert vcfheader.report_fields(["SNVDP", "GL", "AOPSUM", "AFP", "INFO/AFPRIOR"]) == (["AFPRIOR", "AFP", "AOPSUM", "SNVDP"], ["AFP", "GL", "SNVDP"])
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


def test_every_flag_of_the_reference_programs_is_accepted_with_its_arity_and_default():
    """tests/golden/cli_flags.json (made by tests/golden/make_cli_flags.py from application/arguments.py:742-838) lists the
    flags of the reference's three programs: each must parse here with the same number of values, the same default and
    the same type -- no flag is silently dropped, none is missing."""
    import json

    from mchap_amd import cli

    table = json.load(open(os.path.join(os.path.dirname(HERE), "cli_flags.json")))
    assert set(table) == {"assemble", "call", "call-exact"}
    for program, rows in table.items():
        parser = cli.build_parser(program)
        defaults = vars(parser.parse_args([]))
        assert len(rows) >= 16
        for row in rows:
            dest = row.get("dest") or row["flag"].lstrip("-").replace("-", "_")
            assert dest in defaults, "%s: %s is not accepted" % (program, row["flag"])
            if row["kind"] == "flag":
                assert defaults[dest] == (row["action"] == "store_false")
                assert vars(parser.parse_args([row["flag"]]))[dest] == (row["action"] == "store_true")
                continue
            assert defaults[dest] == row["default"], (program, row["flag"], defaults[dest], row["default"])
            n = row["nargs"]
            sample = {"int": "3", "float": "0.5", "str": "x"}[row["type"]]
            values = [sample] * (n if isinstance(n, int) else 2)
            got = vars(parser.parse_args([row["flag"]] + values))[dest]
            cast = {"int": int, "float": float, "str": str}[row["type"]]
            assert got == [cast(v) for v in values], (program, row["flag"], got)
            if isinstance(n, int):
                with pytest.raises(SystemExit):  # the wrong number of values is refused, as by the reference's parser
                    parser.parse_args([row["flag"]] + values + [sample] + (["--bam"] if False else []))


def test_pools_temperatures_region_and_reference_index(tmp_path):
    from mchap_amd import application, io

    bams = {"A": "a.bam", "B": "b.bam", "C": "c.bam"}
    assert io.sample_pools(bams, None) == {"A": [("A", "a.bam")], "B": [("B", "b.bam")], "C": [("C", "c.bam")]}
    assert io.sample_pools(bams, "POOL") == {"POOL": [("A", "a.bam"), ("B", "b.bam"), ("C", "c.bam")]}
    f = tmp_path / "pools.txt"
    f.write_text("A\tP1\nB\tP1\nC\tP2\nA\tP2\n")   # a sample may be in several pools (arguments.py:848-887)
    assert io.sample_pools(bams, str(f)) == {"P1": [("A", "a.bam"), ("B", "b.bam")], "P2": [("C", "c.bam"), ("A", "a.bam")]}
    f.write_text("A\tP1\nB\tP1\n")
    with pytest.raises(ValueError, match="not been assigned"):
        io.sample_pools(bams, str(f))
    f.write_text("A\tP1\nB\tP1\nC\tP1\nD\tP1\n")
    with pytest.raises(ValueError, match="do not match"):
        io.sample_pools(bams, str(f))
    # --mcmc-temperatures: a ladder for all, or a per-sample file; sorted, 1.0 appended (arguments.py:1122-1166)
    assert io.sample_temperatures(["1.0"], ["A", "B"]) == {"A": (1.0,), "B": (1.0,)}
    assert io.sample_temperatures(["0.5", "0.1"], ["A"]) == {"A": (0.1, 0.5, 1.0)}
    t = tmp_path / "temps.txt"
    t.write_text("B\t0.2\t0.05\n")
    assert io.sample_temperatures([str(t)], ["A", "B"]) == {"A": (1.0,), "B": (0.05, 0.2, 1.0)}
    assert io.parse_region("chr1:17590-17709") == ("chr1", 17590, 17709)
    assert application.assemble_targets(None, "chr1:10-20", "x") == [("chr1", 10, 20, "x")]
    with pytest.raises(ValueError):
        application.assemble_targets("a.bed", "chr1:10-20")
    with pytest.raises(ValueError):
        application.assemble_targets(None, None)
    # a reference known by its index only: lengths for the header, N for the unknown bases, REF alleles of the variants kept
    fa = tmp_path / "ref.fa"
    (tmp_path / "ref.fa.fai").write_text("chrZ\t1000\t6\t60\t61\n")
    with pytest.raises(IOError):
        io.Reference(str(fa))  # index-only references are opt-in
    ref = io.Reference(str(fa), allow_index_only=True)
    assert ref.contigs == [("chrZ", 1000)] and ref.fetch("chrZ", 10, 16) == "NNNNNN"
    recs = [dict(chrom="chrZ", pos=12, id=".", ref="A", alts=("C",), info={}), dict(chrom="chrZ", pos=15, id=".", ref="G", alts=("T", "A"), info={})]
    locus = io.DenovoLocus("chrZ", 10, 16, "t", recs, ref.fetch("chrZ", 10, 16), sequence_known=ref.known)
    assert io.DenovoLocus("chrZ", 10, 16, "t", recs, "NNNNNN").sequence == "NNNNNN"  # a real FASTA's N bases stay
    assert locus.sequence == "NANNGN" and locus.format_haplotype([1, 2]) == "NCNNAN" and locus.n_alleles == [2, 3]
    with pytest.raises(IOError):
        io.Reference(str(tmp_path / "other.fa"))


def test_gzip_vcf_input_and_read_filters(tmp_path):
    from mchap_amd import application, io

    src = os.path.join(HERE, "simple.vcf")
    gz = tmp_path / "simple.vcf.gz"
    with gzip.open(gz, "wt") as f:
        f.write(open(src).read())
    assert io.read_vcf(str(gz)) == io.read_vcf(src)
    assert io.vcf_contigs(str(gz)) == io.vcf_contigs(src)
    # --mapping-quality / --keep-*-reads reach extract_read_variants (io/bam.py:54-229)
    bam = io.read_alignments(os.path.join(HERE, "simple.sample1.bam"))
    bed = io.read_bed4(os.path.join(HERE, "simple.bed"))
    _, variants = io.read_vcf(src)
    contig, start, stop, name = bed[0]
    locus = io.DenovoLocus(contig, start, stop, name, variants, "A" * (stop - start))
    n_default = len(io.extract_read_variants(locus, bam, "SAMPLE1")[0])
    mapqs = sorted({r["mapq"] for r in bam[2]})
    assert len(io.extract_read_variants(locus, bam, "SAMPLE1", min_quality=max(mapqs) + 1)[0]) == 0
    assert len(io.extract_read_variants(locus, bam, "SAMPLE1", min_quality=0, skip_duplicates=False, skip_qcfail=False, skip_supplementary=False)[0]) >= n_default
    src_ = application.ReadSource({"SAMPLE1": os.path.join(HERE, "simple.sample1.bam")}, mapping_quality=max(mapqs) + 1)
    assert len(src_.reads(locus, "SAMPLE1")["calls"]) == 0
    # read-group field "ID": the groups' own ids name the samples
    rg_ids = list(io.bam_header(os.path.join(HERE, "simple.sample1.bam"))[1])
    assert list(io.sample_bam_table([os.path.join(HERE, "simple.sample1.bam")], "ID")) == rg_ids
