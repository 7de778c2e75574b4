"""Measurement aid: cProfile of one `assemble` job on the block path (host functions by own time).  Usage: python tools/e2e_profile.py [loci]"""
import cProfile, os, pstats, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mchap_amd import application, io, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = tempfile.mkdtemp(prefix="mchap_e2e_")
try:
    job = synth.synth_assembly_inputs(d, n_loci=n, n_samples=1, reads_per_locus=100)

    def once():
        source = application.ReadSource(io.sample_bam_table(job["bams"]), workers=4)
        return list(application.assemble(job["bed"], job["vcf"], io.Reference(job["fasta"]), source, ploidy=4, steps=2000, burn=1000, chains=2, seed=42))

    once()
    pr = cProfile.Profile()
    pr.enable()
    once()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
finally:
    shutil.rmtree(d, ignore_errors=True)
