import cProfile, pstats, io as _io, os, sys, tempfile, shutil
sys.path.insert(0, "/root/repo")
from mchap_amd import cli, synth
d = tempfile.mkdtemp(prefix="mchap_call_prof_")
job = synth.synth_assembly_inputs(d, n_loci=300, n_samples=4, reads_per_locus=60)
vcf = os.path.join(d, "haplotypes.vcf")
out = _io.StringIO()
cli.run(["mchap_amd", "assemble", "--bam"] + job["bams"] + ["--targets", job["bed"], "--variants", job["vcf"], "--reference", job["fasta"], "--ploidy", "4"], out)
open(vcf, "w").write(out.getvalue())
argv = ["mchap_amd", "call", "--bam"] + job["bams"] + ["--haplotypes", vcf, "--ploidy", "4"]
cli.run(argv, _io.StringIO())
pr = cProfile.Profile(); pr.enable(); cli.run(argv, _io.StringIO()); pr.disable()
s = _io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
shutil.rmtree(d, ignore_errors=True)
