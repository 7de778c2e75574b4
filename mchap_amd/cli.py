"""Command line of the programs built on the kernels: `python -m mchap_amd {assemble,call,call-exact} ...` with the
reference's flag names, defaults and meaning (application/cli.py:14-60; application/arguments.py: the argument tables
ASSEMBLE_MCMC_PARSER_ARGUMENTS / CALL_EXACT_PARSER_ARGUMENTS / CALL_MCMC_PARSER_ARGUMENTS and the collect_* functions)
for these three programs: every flag those tables define is accepted here with the same arity and default.  Output: a VCF
on standard output (header from mchap_amd.vcfheader, records from mchap_amd.application).

Differences, by design: `--cores N` sizes the pool of host threads that inflate and parse the alignment files (the
reference forks N processes over blocks of loci; here a block of loci is one GPU launch); with several ranks
(`python -m torch.distributed.run --nproc-per-node N -m mchap_amd ...`, one rank per GPU) the targets / records are
sharded over the ranks and rank 0 writes the one VCF; `--reference` may name a FASTA whose file is absent when its
`.fai` index is present (contig lengths for the header; unknown reference bases are written as N)."""
import argparse
import sys

from . import __version__

PROGRAMS = ("assemble", "call", "call-exact")


def _flag(p, name, dest, action, help):
    p.add_argument(name, dest=dest, action=action, default=(action == "store_false"), help=help)


def _sample_args(p, dirmul_nargs):
    p.add_argument("--bam", type=str, nargs="+", default=[],
                   help="BAM / SAM file(s), a text file with one path per line, or a text file of sample<TAB>path lines")
    p.add_argument("--ploidy", type=str, nargs=1, default=["2"], help="ploidy of all samples, or a file of sample<TAB>ploidy lines")
    if dirmul_nargs == 2:
        p.add_argument("--use-dirmul-prior", type=str, nargs=2, default=[None, None],
                       help="inbreeding (a value or a file of sample<TAB>value lines) and the INFO field of prior allele frequencies")
    else:
        p.add_argument("--use-dirmul-prior", type=str, nargs=1, default=[None],
                       help="Dirichlet-multinomial prior over all SNV combinations: inbreeding as a value or a file of sample<TAB>value lines")
    p.add_argument("--sample-pool", type=str, nargs=1, default=[None],
                   help="pool samples into one genotype: the name of a single pool of all samples, or a file of sample<TAB>pool lines")


def _read_args(p):
    p.add_argument("--base-error-rate", type=float, nargs=1, default=[0.0024])
    _flag(p, "--use-base-phred-scores", "ignore_base_phred_scores", "store_false", "use the base phred scores as a source of error")
    p.add_argument("--mapping-quality", type=int, nargs=1, default=[20], help="minimum mapping quality of the reads used")
    _flag(p, "--keep-duplicate-reads", "skip_duplicates", "store_false", "use reads marked as duplicates")
    _flag(p, "--keep-qcfail-reads", "skip_qcfail", "store_false", "use reads marked as qcfail")
    _flag(p, "--keep-supplementary-reads", "skip_supplementary", "store_false", "use reads marked as supplementary")
    p.add_argument("--read-group-field", type=str, nargs=1, default=["SM"], help='read-group field used as the sample id: "SM" or "ID"')


def _mcmc_args(p):
    p.add_argument("--mcmc-chains", type=int, nargs=1, default=[2])
    p.add_argument("--mcmc-steps", type=int, nargs=1, default=[2000])
    p.add_argument("--mcmc-burn", type=int, nargs=1, default=[1000])
    p.add_argument("--mcmc-seed", type=int, nargs=1, default=[42])
    p.add_argument("--mcmc-chain-incongruence-threshold", type=float, nargs=1, default=[0.60])


def build_parser(program):
    """The parser of one program: the flags of the reference's argument table for it (application/arguments.py:742-838)."""
    p = argparse.ArgumentParser("mchap_amd " + program)
    if program == "assemble":
        _sample_args(p, 1)
        p.add_argument("--reference", type=str, nargs=1, default=[None], help="reference FASTA")
        p.add_argument("--reference-index-only", action="store_true",
                       help="(not a reference flag) accept a --reference of which only the .fai index exists: N for unknown bases")
        p.add_argument("--region", type=str, nargs=1, default=[None], help="a single target contig:start-stop (not with --targets)")
        p.add_argument("--region-id", type=str, nargs=1, default=[None])
        p.add_argument("--targets", type=str, nargs=1, default=[None], help="BED4 file of target loci")
        p.add_argument("--variants", type=str, nargs=1, default=[None], help="VCF (text or bgzip) of the SNVs to assemble over")
        _read_args(p)
        _mcmc_args(p)
        p.add_argument("--mcmc-fix-homozygous", type=float, nargs=1, default=[0.999])
        p.add_argument("--mcmc-llk-cache-threshold", type=int, nargs=1, default=[100])
        p.add_argument("--mcmc-recombination-step-probability", type=float, nargs=1, default=[0.5])
        p.add_argument("--mcmc-dosage-step-probability", type=float, nargs=1, default=[1.0])
        p.add_argument("--mcmc-partial-dosage-step-probability", type=float, nargs=1, default=[0.5])
        p.add_argument("--mcmc-temperatures", type=str, nargs="*", default=["1.0"],
                       help="inverse temperatures of parallel tempered chains, or a file of sample<TAB>t1<TAB>t2... lines")
        p.add_argument("--haplotype-posterior-threshold", type=float, nargs=1, default=[0.20])
    else:
        _sample_args(p, 2)
        p.add_argument("--reference", type=str, nargs=1, default=[None], help="reference FASTA (only needed for CRAM input: not read)")
        p.add_argument("--haplotypes", type=str, nargs=1, default=[None], help="VCF (text or bgzip) of known haplotypes")
        p.add_argument("--filter-input-haplotypes", type=str, nargs=1, default=[None],
                       help="'<field><operator><value>': INFO field of Number A or R, one of = > < >= <= !=, a number")
        p.add_argument("--prior-frequencies", type=str, nargs=1, default=[None],
                       help="INFO field of prior allele frequencies (the second value of --use-dirmul-prior)")
        _read_args(p)
        if program == "call":
            _mcmc_args(p)
    p.add_argument("--report", type=str, nargs="*", default=[],
                   help="extra fields: AFPRIOR ACP AFP AOP AOPSUM GP GL SNVDP, optionally with an INFO/ or FORMAT/ prefix")
    p.add_argument("--cores", type=int, nargs=1, default=[1], help="host threads reading the alignment files")
    return p


def _ranks():
    """(rank, world, dist) when launched under torch.distributed.run with more than one rank, else (0, 1, None)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    if not dist.is_initialized():
        # RCCL needs one GPU per rank; MCHAP_DIST_BACKEND=gloo runs several ranks on one GPU (how it is tested)
        dist.init_process_group(backend=os.environ.get("MCHAP_DIST_BACKEND", "nccl" if torch.cuda.device_count() >= world else "gloo"))
    return rank, world, dist


def run(argv, out=None):
    """argv as sys.argv (argv[1] names the program).  Writes the VCF to `out` (default stdout; with several ranks: rank 0
    only); returns the number of records this process formatted."""
    from . import application, io, vcfheader

    out = out or sys.stdout
    program = argv[1]
    args = build_parser(program).parse_args(argv[2:])
    # must have some source of error in reads (application/arguments.py:1190-1195)
    if args.ignore_base_phred_scores and args.base_error_rate[0] == 0.0:
        raise ValueError("Cannot ignore base phred scores if --base-error-rate is 0")
    id_field = args.read_group_field[0]
    if id_field not in ("SM", "ID"):
        raise ValueError('--read-group-field must be "SM" or "ID"')
    sample_bams = io.sample_pools(io.sample_bam_table(args.bam, id_field), args.sample_pool[0])
    samples = list(sample_bams)
    ploidy = io.sample_values(args.ploidy[0], samples, int)
    report = list(args.report)
    vcfheader.report_fields(report)  # (unknown names fail here, before any work)
    rank, world, dist = _ranks()
    shard = (rank, world) if world > 1 else None
    source = application.ReadSource(sample_bams, error_rate=args.base_error_rate[0], use_phred=not args.ignore_base_phred_scores,
                                    read_group_field=id_field, mapping_quality=args.mapping_quality[0],
                                    skip_duplicates=args.skip_duplicates, skip_qcfail=args.skip_qcfail,
                                    skip_supplementary=args.skip_supplementary, workers=args.cores[0])
    seed = None
    if program == "assemble":
        if args.targets[0] is not None and args.region[0] is not None:
            raise ValueError("Cannot combine --targets and --region arguments.")
        if args.variants[0] is None or args.reference[0] is None:
            raise ValueError("--variants and --reference are required")
        seed = args.mcmc_seed[0]
        inbreeding = io.sample_values(args.use_dirmul_prior[0], samples, float)
        reference = io.Reference(args.reference[0], allow_index_only=args.reference_index_only)
        contigs = reference.contigs
        targets = application.assemble_targets(args.targets[0], args.region[0], args.region_id[0])
        records = application.assemble(
            None, args.variants[0], reference, source, ploidy=ploidy, inbreeding=inbreeding, steps=args.mcmc_steps[0],
            burn=args.mcmc_burn[0], chains=args.mcmc_chains[0], seed=seed, haplotype_posterior_threshold=args.haplotype_posterior_threshold[0],
            incongruence_threshold=args.mcmc_chain_incongruence_threshold[0], report=report,
            temperatures=io.sample_temperatures(args.mcmc_temperatures, samples), targets=targets, shard=shard,
            fix_homozygous=args.mcmc_fix_homozygous[0], recombination_step_probability=args.mcmc_recombination_step_probability[0],
            partial_dosage_step_probability=args.mcmc_partial_dosage_step_probability[0],
            dosage_step_probability=args.mcmc_dosage_step_probability[0], llk_cache_threshold=args.mcmc_llk_cache_threshold[0])
    else:
        if args.haplotypes[0] is None:
            raise ValueError("--haplotypes is required")
        inb_arg, tag = args.use_dirmul_prior
        tag = tag or args.prior_frequencies[0]
        inbreeding = io.sample_values(inb_arg, samples, float)
        contigs = io.vcf_contigs(args.haplotypes[0]) or (io.bam_header(next(iter(source.bams)))[0] if source.bams else [])
        if program == "call-exact":
            records = application.call_exact(args.haplotypes[0], source, ploidy=ploidy, report=report, prior_frequencies_tag=tag,
                                             inbreeding=inbreeding, filter_input_haplotypes=args.filter_input_haplotypes[0], shard=shard)
        else:
            seed = args.mcmc_seed[0]
            records = application.call(args.haplotypes[0], source, ploidy=ploidy, report=report, prior_frequencies_tag=tag,
                                       inbreeding=inbreeding, steps=args.mcmc_steps[0], burn=args.mcmc_burn[0], chains=args.mcmc_chains[0],
                                       seed=seed, incongruence_threshold=args.mcmc_chain_incongruence_threshold[0],
                                       filter_input_haplotypes=args.filter_input_haplotypes[0], shard=shard)
    lines = list(records) if world > 1 else records
    if world > 1:
        # the only exchange: every rank's formatted record lines to rank 0, which writes them in target order (the shards
        # are contiguous, so rank order is target order)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(lines, gathered, dst=0)
        n_mine = len(lines)
        if rank != 0:
            return n_mine
        lines = [ln for part in gathered for ln in part]
    for line in vcfheader.header_lines(program, ["mchap_amd"] + list(argv[1:]), samples, contigs, report=report, random_seed=seed):
        out.write(line + "\n")
    n = 0
    run.limit_records = 0
    for line in lines:
        out.write(line + "\n")
        n += 1
        if program == "assemble" and line.split("\t", 7)[6] == "LIMIT":
            run.limit_records += 1  # (a target beyond the build's shape limits: written with null genotypes, and main() says so)
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
    if getattr(run, "limit_records", 0):
        sys.stderr.write("mchap_amd: %d target(s) were not assembled (FILTER=LIMIT): exit status 3\n" % run.limit_records)
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
