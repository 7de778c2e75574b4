"""Measurement aid: `mchap call` end to end on a synthetic job -- assemble writes the haplotype VCF of `loci` targets x `samples`
samples, `call` re-calls every sample against it (files -> VCF records through application.call) -- with the traces summarised on
the device (CallingMCMC.fit_batch_summaries, round 5) and, for comparison, by the host classes on downloaded traces as before.
Usage: python tools/call_e2e_once.py [loci] [samples]"""
import io as _io, os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mchap_amd import cli, synth
from mchap_amd.calling_mcmc import CallingMCMC, CallSummary

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = tempfile.mkdtemp(prefix="mchap_call_e2e_")
try:
    job = synth.synth_assembly_inputs(d, n_loci=n, n_samples=ns, reads_per_locus=60)
    vcf = os.path.join(d, "haplotypes.vcf")
    out = _io.StringIO()
    cli.run(["mchap_amd", "assemble", "--bam"] + job["bams"] + ["--targets", job["bed"], "--variants", job["vcf"], "--reference", job["fasta"], "--ploidy", "4"], out)
    open(vcf, "w").write(out.getvalue())
    argv = ["mchap_amd", "call", "--bam"] + job["bams"] + ["--haplotypes", vcf, "--ploidy", "4"]
    device = CallingMCMC.start_batch_summaries

    def by_host(self, reads, read_counts=None, initial=None, haplotypes=None, prior=None, stream_ids=None, burn=0, incongruence_threshold=0.6, max_states=512,
                stream=None):
        return dict(done=[CallSummary.of_trace(t.burn(burn), incongruence_threshold) for t in self.fit_batch(reads, read_counts, initial, haplotypes, prior, stream_ids)])

    texts = {}
    for name, fn in (("device summaries", device), ("host classes", by_host), ("device summaries", device)):
        CallingMCMC.start_batch_summaries = fn
        o = _io.StringIO()
        t0 = time.perf_counter()
        cli.run(argv, o)
        dt = time.perf_counter() - t0
        recs = [l for l in o.getvalue().splitlines() if not l.startswith("#")]
        texts[name] = recs
        print("%-17s %d records x %d samples  %.1f ms  %.0f units/s" % (name, len(recs), ns, dt * 1e3, len(recs) * ns / dt), flush=True)
    print("same records:", texts["device summaries"] == texts["host classes"])
finally:
    shutil.rmtree(d, ignore_errors=True)
