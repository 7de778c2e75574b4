"""Measurement aid: bench.py's extra.program_e2e (files -> VCF records through application.assemble) alone, block path and
per-locus path.  Usage: python tools/e2e_once.py [loci] [reps]"""
import json, os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mchap_amd import application, io, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
upb = int(sys.argv[3]) if len(sys.argv) > 3 else None   # units per block (default: by free memory)
d = tempfile.mkdtemp(prefix="mchap_e2e_")
try:
    job = synth.synth_assembly_inputs(d, n_loci=n, n_samples=1, reads_per_locus=100)
    for bp in ((None, False) if upb is None else (None,)):
        for r in range(reps):
            tm = {}
            t0 = time.perf_counter()
            source = application.ReadSource(io.sample_bam_table(job["bams"]), workers=4)
            ref = io.Reference(job["fasta"])
            lines = list(application.assemble(job["bed"], job["vcf"], ref, source, ploidy=4, steps=2000, burn=1000, chains=2, seed=42, timings=tm, block_path=bp, units_per_block=upb))
            dt = time.perf_counter() - t0
            print("block_path=%s  %d loci  %.1f ms  %.0f loci/s  %s" % (bp, len(lines), dt * 1e3, len(lines) / dt,
                                                                     {k: round(v * 1e3, 1) for k, v in tm.items() if k.endswith("_s")}), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
