"""VCF meta-information of the programs' output: the header block `mchap assemble / call / call-exact` write before
their records (reference io/vcf/headermeta.py, infofields.py, formatfields.py, filters.py, application/baseclass.py:
392-434).  The field identifiers, Number / Type and description strings are the output format itself (a consumer of
the reference's VCFs must find the same declarations), so they are tabulated here once."""
from datetime import date



FILTERS = [
    ("PASS", "All filters passed"),
    ("NOA", "No observed alleles at locus"),
    ("AF0", "All alleles have prior allele frequency of zero"),
]

# `assemble` only, not a reference filter: a target beyond the library's shape limits (README: SNVs per target, bits of sampled
# alleles per haplotype) is written with null genotypes and this filter instead of being left out of the file
LIMIT_FILTER = ("LIMIT", "Target not assembled: beyond the shape limits of this build (record kept with null genotypes)")
# (the call programs: a record whose units the exact caller / call sampler does not take -- more than 2^62 genotypes, more than
# 256 known haplotypes for the sampler, tables beyond the LDS -- is written with null genotypes and this filter, round 5)
LIMIT_FILTER_CALL = ("LIMIT", "Record not called: beyond the shape limits of this build (record kept with null genotypes)")

# (id, Number, Type, Description); the order is the order of the header and of the INFO column
INFO_FIELDS = [
    ("AN", "1", "Integer", "Total number of alleles in called genotypes"),
    ("UAN", "1", "Integer", "Total number of unique alleles in called genotypes"),
    ("AC", "A", "Integer", "Allele count in genotypes, for each ALT allele, in the same order as listed"),
    ("REFMASKED", "0", "Flag", "Reference allele is masked"),
    ("NS", "1", "Integer", "Number of samples with data"),
    ("MCI", "1", "Integer", "Number of samples with incongruent Markov chain replicates"),
    ("DP", "1", "Integer", "Combined depth across samples"),
    ("RCOUNT", "1", "Integer", "Total number of observed reads across all samples"),
    ("END", "1", "Integer", "End position on CHROM"),
    ("NVAR", "1", "Integer", "Number of input variants within assembly locus"),
    ("SNVPOS", ".", "Integer", "Relative (1-based) positions of SNVs within haplotypes"),
]
OPTIONAL_INFO_FIELDS = {
    "AFPRIOR": ("R", "Float", "Prior allele frequencies"),
    "ACP": ("R", "Float", "Posterior allele counts"),
    "AFP": ("R", "Float", "Posterior mean allele frequencies"),
    "AOP": ("R", "Float", "Posterior probability of allele occurring across all samples"),
    "AOPSUM": ("R", "Float", "Posterior estimate of the number of samples containing an allele"),
    "SNVDP": (".", "Integer", "Read depth at each SNV position"),
}
FORMAT_FIELDS = [
    ("GT", "1", "String", "Genotype"),
    ("GQ", "1", "Integer", "Genotype quality"),
    ("SQ", "1", "Integer", "Genotype support quality"),
    ("DP", "1", "Integer", "Read depth"),
    ("RCOUNT", "1", "Integer", "Total count of read pairs within haplotype interval"),
    ("RCALLS", "1", "Integer", "Total count of read base calls matching a known variant"),
    ("MEC", "1", "Integer", "Minimum error correction"),
    ("MECP", "1", "Float", "Minimum error correction proportion"),
    ("GPM", "1", "Float", "Genotype posterior mode probability"),
    ("SPM", "1", "Float", "Genotype support posterior mode probability"),
    ("MCI", "1", "Integer", "Replicate Markov-chain incongruence, 0 = none, 1 = incongruence, 2 = putative CNV"),
]
OPTIONAL_FORMAT_FIELDS = {
    "ACP": ("R", "Float", "Posterior allele counts"),
    "AFP": ("R", "Float", "Posterior mean allele frequencies"),
    "AOP": ("R", "Float", "Posterior probability of allele occurring"),
    "GP": ("G", "Float", "Genotype posterior probabilities"),
    "GL": ("G", "Float", "Genotype likelihoods"),
    "SNVDP": (".", "Integer", "Read depth at each SNV position"),
}


def report_fields(report):
    """--report arguments -> (optional INFO ids, optional FORMAT ids).  A bare name asks for both variants of the field; an
    `INFO/` or `FORMAT/` prefix for one.  Whatever the order of the arguments, the fields come out in the order of the
    reference's field tables (io/vcf/infofields.py:125, formatfields.py:157; application/arguments.py:1169-1185) -- which is
    the order of the header lines and of the INFO / FORMAT columns.  Unknown names are an error (the reference ignores them)."""
    asked = set(report or ())
    for name in asked:
        base = name.split("/", 1)[1] if name.startswith(("INFO/", "FORMAT/")) else name
        ok = (base in OPTIONAL_INFO_FIELDS and not name.startswith("FORMAT/")) or (base in OPTIONAL_FORMAT_FIELDS and not name.startswith("INFO/"))
        if not ok:
            raise ValueError("Unknown %s field to report: %s" % ("INFO" if name.startswith("INFO/") else "FORMAT" if name.startswith("FORMAT/") else "INFO/FORMAT", name))
    info = [f for f in OPTIONAL_INFO_FIELDS if f in asked or "INFO/" + f in asked]
    fmt = [f for f in OPTIONAL_FORMAT_FIELDS if f in asked or "FORMAT/" + f in asked]
    return info, fmt


def header_lines(program, command, samples, contigs, report=(), random_seed=None, today=None, version=None):
    """The header block as a list of lines.  contigs: [(name, length)]; command: the argv list or a string."""
    from . import __version__

    d = today or date.today()
    cmd = command if isinstance(command, str) else '"%s"' % " ".join(command)
    out = ["##fileformat=VCFv4.3", "##fileDate=%04d%02d%02d" % (d.year, d.month, d.day),
           "##source=mchap_amd v%s (%s)" % (version or __version__, program), "##phasing=None", "##commandline=%s" % cmd,
           "##randomseed=%s" % random_seed]
    out += ["##contig=<ID=%s,length=%d>" % (n, l) for n, l in contigs]
    out += ['##FILTER=<ID=%s,Description="%s">' % f for f in FILTERS + ([LIMIT_FILTER] if program == "assemble" else [LIMIT_FILTER_CALL])]
    info_opt, fmt_opt = report_fields(report)
    for fid, num, typ, descr in INFO_FIELDS:
        out.append('##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">' % (fid, num, typ, descr))
    for fid in info_opt:
        out.append('##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">' % ((fid,) + OPTIONAL_INFO_FIELDS[fid]))
    for fid, num, typ, descr in FORMAT_FIELDS:
        out.append('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="%s">' % (fid, num, typ, descr))
    for fid in fmt_opt:
        out.append('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="%s">' % ((fid,) + OPTIONAL_FORMAT_FIELDS[fid]))
    out.append("#" + "\t".join(["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + list(samples)))
    return out
