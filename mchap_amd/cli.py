"""Command line of the programs built on the kernels: `python -m mchap_amd {assemble,call,call-exact} ...` with the
reference's flag names and meaning (application/cli.py:14-60, application/arguments.py) for the flags that concern these
three programs.  Output: a VCF on standard output (header from mchap_amd.vcfheader, records from mchap_amd.application).
Not restated: --region / --sample-pool / --sample-parents (pedigree), read-group field selection, --cores (the batch is
one GPU launch; multi-GPU sharding is by target list, see mchap_amd/shard.py)."""
import argparse
import sys

from . import __version__

PROGRAMS = ("assemble", "call", "call-exact")


def _common(p):
    p.add_argument("--bam", type=str, nargs="+", default=[], help="BAM file(s), a file of BAM paths, or a file of sample<TAB>path lines")
    p.add_argument("--ploidy", type=str, nargs=1, default=["2"], help="ploidy of all samples, or a file of sample<TAB>ploidy lines")
    p.add_argument("--base-error-rate", type=float, nargs=1, default=[0.0024])
    p.add_argument("--use-base-phred-scores", action="store_true", default=False)
    p.add_argument("--report", type=str, nargs="*", default=[], help="extra INFO/FORMAT fields: AFPRIOR ACP AFP AOP GP GL SNVDP")
    p.add_argument("--mapping-quality", type=int, nargs=1, default=[20])
    p.add_argument("--mcmc-seed", type=int, nargs=1, default=[None])
    p.add_argument("--mcmc-chains", type=int, nargs=1, default=[2])
    p.add_argument("--mcmc-chain-incongruence-threshold", type=float, nargs=1, default=[0.60])


def build_parser(program):
    p = argparse.ArgumentParser("mchap_amd " + program)
    if program == "assemble":
        p.add_argument("--targets", type=str, nargs=1, required=True, help="BED4 file of target loci")
        p.add_argument("--variants", type=str, nargs=1, required=True, help="VCF file of SNVs")
        p.add_argument("--reference", type=str, nargs=1, required=True, help="reference FASTA")
        _common(p)
        p.add_argument("--use-dirmul-prior", type=str, nargs=1, default=[None], help="inbreeding value or sample<TAB>value file")
        p.add_argument("--haplotype-posterior-threshold", type=float, nargs=1, default=[0.20])
        p.add_argument("--mcmc-steps", type=int, nargs=1, default=[2000])
        p.add_argument("--mcmc-burn", type=int, nargs=1, default=[1000])
        p.add_argument("--mcmc-temperatures", type=float, nargs="*", default=[1.0])
        p.add_argument("--mcmc-fix-homozygous", type=float, nargs=1, default=[0.999])
        p.add_argument("--mcmc-recombination-step-probability", type=float, nargs=1, default=[0.5])
        p.add_argument("--mcmc-partial-dosage-step-probability", type=float, nargs=1, default=[0.5])
        p.add_argument("--mcmc-dosage-step-probability", type=float, nargs=1, default=[1.0])
        p.add_argument("--mcmc-llk-cache-threshold", type=int, nargs=1, default=[100])
    else:
        p.add_argument("--haplotypes", type=str, nargs=1, required=True, help="VCF file of known haplotypes")
        _common(p)
        p.add_argument("--use-dirmul-prior", type=str, nargs=2, default=[None, None],
                       help="inbreeding (value or file) and the INFO field of prior allele frequencies")
        p.add_argument("--prior-frequencies", type=str, nargs=1, default=[None])
        p.add_argument("--filter-input-haplotypes", type=str, nargs=1, default=[None],
                       help="'<field><operator><value>': INFO field of Number A or R, one of = > < >= <= !=, a number")
        if program == "call":
            p.add_argument("--mcmc-steps", type=int, nargs=1, default=[2000])
            p.add_argument("--mcmc-burn", type=int, nargs=1, default=[1000])
    return p


def run(argv, out=None):
    """argv as sys.argv (argv[1] names the program).  Writes the VCF to `out` (default stdout); returns the number of records."""
    from . import application, io, vcfheader

    out = out or sys.stdout
    program = argv[1]
    args = build_parser(program).parse_args(argv[2:])
    sample_bams = io.sample_bam_table(args.bam)
    samples = list(sample_bams)
    ploidy = io.sample_values(args.ploidy[0], samples, int)
    report = list(args.report)
    seed = args.mcmc_seed[0]
    contigs = io.bam_header(next(iter(sample_bams.values())))[0] if sample_bams else []
    if program == "assemble":
        inbreeding = io.sample_values(args.use_dirmul_prior[0], samples, float)
        reference = io.read_fasta(args.reference[0])
        contigs = [(n, len(s)) for n, s in reference.items()]
        records = application.assemble(
            args.targets[0], args.variants[0], reference, sample_bams, ploidy=ploidy, inbreeding=inbreeding, steps=args.mcmc_steps[0],
            burn=args.mcmc_burn[0], chains=args.mcmc_chains[0], seed=seed, error_rate=args.base_error_rate[0],
            use_phred=args.use_base_phred_scores, haplotype_posterior_threshold=args.haplotype_posterior_threshold[0],
            incongruence_threshold=args.mcmc_chain_incongruence_threshold[0], temperatures=tuple(args.mcmc_temperatures),
            fix_homozygous=args.mcmc_fix_homozygous[0], recombination_step_probability=args.mcmc_recombination_step_probability[0],
            partial_dosage_step_probability=args.mcmc_partial_dosage_step_probability[0],
            dosage_step_probability=args.mcmc_dosage_step_probability[0], llk_cache_threshold=args.mcmc_llk_cache_threshold[0])
        report = []
    else:
        inb_arg, tag = args.use_dirmul_prior
        tag = tag or args.prior_frequencies[0]
        inbreeding = io.sample_values(inb_arg, samples, float)
        if program == "call-exact":
            records = application.call_exact(args.haplotypes[0], sample_bams, ploidy=ploidy, report=report,
                                             base_error_rate=args.base_error_rate[0], use_base_phred_scores=args.use_base_phred_scores,
                                             prior_frequencies_tag=tag, inbreeding=inbreeding,
                                             filter_input_haplotypes=args.filter_input_haplotypes[0])
        else:
            records = application.call(args.haplotypes[0], sample_bams, ploidy=ploidy, report=report,
                                       base_error_rate=args.base_error_rate[0], use_base_phred_scores=args.use_base_phred_scores,
                                       prior_frequencies_tag=tag, inbreeding=inbreeding, steps=args.mcmc_steps[0], burn=args.mcmc_burn[0],
                                       chains=args.mcmc_chains[0], seed=seed,
                                       incongruence_threshold=args.mcmc_chain_incongruence_threshold[0],
                                       filter_input_haplotypes=args.filter_input_haplotypes[0])
    for line in vcfheader.header_lines(program, ["mchap_amd"] + list(argv[1:]), samples, contigs, report=report, random_seed=seed):
        out.write(line + "\n")
    n = 0
    for line in records:
        out.write(line + "\n")
        n += 1
    return n


def main(argv=None):
    argv = sys.argv if argv is None else argv
    parser = argparse.ArgumentParser("MI355X-native MCHap programs")
    parser.add_argument("-v", "--version", action="version", version="mchap_amd %s" % __version__)
    parser.add_argument("program", nargs=1, choices=PROGRAMS, help="sub-program")
    if len(argv) < 2:
        parser.print_help()
        return 0
    parser.parse_args(argv[1:2])
    run(argv)
    return 0


if __name__ == "__main__":
    sys.exit(main())
