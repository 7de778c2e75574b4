"""GPU: the programs end to end through the command-line shell (python -m mchap_amd ...): VCF header + records from BAM /
VCF / BED / FASTA files, against the reference's golden VCFs (call-exact, deep assemble) and, for `mchap call` (whose
goldens depend on numba's generator), against call-exact on the same data."""
import io as _io
import os

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")
MIXED = ["simple.sample1.bam", "simple.sample2.deep.bam", "simple.sample3.bam"]
DEEP = ["simple.sample1.deep.bam", "simple.sample2.deep.bam", "simple.sample3.deep.bam"]


def _records(text):
    """the VCF record lines of a program's output (other lines -- e.g. a launcher's or gloo's chatter on a shared stdout -- have no
    tab-separated columns)"""
    return [ln for ln in text.splitlines() if ln and not ln.startswith("#") and ln.count("\t") >= 9]


def _golden(name):
    return [ln.rstrip("\n") for ln in open(os.path.join(HERE, name)) if ln.strip() and not ln.startswith("#")]


def test_call_exact_program_reproduces_the_golden_vcf():
    from mchap_amd import cli

    out = _io.StringIO()
    argv = ["mchap_amd", "call-exact", "--bam"] + [os.path.join(HERE, f) for f in MIXED] + [
        "--ploidy", "4", "--haplotypes", os.path.join(HERE, "mock.input.frequencies.vcf"), "--prior-frequencies", "AFP",
        "--use-dirmul-prior", "0.0", "AFP", "--report", "AFPRIOR", "AFP"]
    n = cli.run(argv, out)
    text = out.getvalue()
    assert _records(text) == _golden("simple.output.mixed_depth.call-exact.frequencies.prior.vcf") and n == len(_records(text))
    head = [ln for ln in text.splitlines() if ln.startswith("#")]
    assert head[0] == "##fileformat=VCFv4.3" and head[-1].split("\t")[9:] == ["SAMPLE1", "SAMPLE2", "SAMPLE3"]
    assert any(ln.startswith("##commandline=") and "call-exact" in ln for ln in head)


def test_assemble_program_reproduces_the_deep_golden_vcf(tmp_path):
    from mchap_amd import cli

    fa = tmp_path / "simple.fasta"
    fa.write_text("".join(">%s\n%s\n" % (c, "A" * 60) for c in ("CHR1", "CHR2", "CHR3")))
    ploidy = tmp_path / "ploidy.txt"
    ploidy.write_text("SAMPLE1\t4\nSAMPLE2\t4\nSAMPLE3\t4\n")
    out = _io.StringIO()
    argv = ["mchap_amd", "assemble", "--bam"] + [os.path.join(HERE, f) for f in DEEP] + [
        "--ploidy", str(ploidy), "--targets", os.path.join(HERE, "simple.bed"), "--variants", os.path.join(HERE, "simple.vcf"),
        "--reference", str(fa), "--use-dirmul-prior", "0.0", "--mcmc-steps", "500", "--mcmc-burn", "100", "--mcmc-seed", "11"]
    cli.run(argv, out)
    assert _records(out.getvalue()) == _golden("simple.output.deep.assemble.vcf")


def test_call_program_agrees_with_call_exact_on_deep_data():
    from mchap_amd import cli

    common = ["--bam"] + [os.path.join(HERE, f) for f in DEEP] + ["--ploidy", "4", "--haplotypes", os.path.join(HERE, "simple.output.deep.assemble.vcf")]
    a, b = _io.StringIO(), _io.StringIO()
    cli.run(["mchap_amd", "call-exact"] + common, a)
    cli.run(["mchap_amd", "call"] + common + ["--mcmc-steps", "600", "--mcmc-burn", "100", "--mcmc-seed", "11", "--report", "AFP"], b)
    ra, rb = _records(a.getvalue()), _records(b.getvalue())
    assert len(ra) == len(rb) > 0
    for x, y in zip(ra, rb):
        fx, fy = x.split("\t"), y.split("\t")
        assert fx[:7] == fy[:7]
        for sx, sy in zip(fx[9:], fy[9:]):
            gx, gy = sx.split(":"), sy.split(":")
            assert gx[0] == gy[0]                      # GT
            assert gy[10] in ("0", "1", "2")           # MCI from the sampler's replicate chains
            assert abs(float(gx[8]) - float(gy[8])) < 0.05 if gx[8] != "." else True   # GPM


def test_call_exact_program_from_sam_text_input():
    """The same program over the reference's SAM renditions of its simple.sample*.bam fixtures (f3: SAM text input)."""
    from mchap_amd import cli

    common = ["--ploidy", "4", "--haplotypes", os.path.join(HERE, "mock.input.frequencies.vcf")]
    a, b = _io.StringIO(), _io.StringIO()
    cli.run(["mchap_amd", "call-exact", "--bam"] + [os.path.join(HERE, "simple.sample%d.bam" % i) for i in (1, 2, 3)] + common, a)
    cli.run(["mchap_amd", "call-exact", "--bam"] + [os.path.join(HERE, "simple.sample%d.sam" % i) for i in (1, 2, 3)] + common, b)
    assert _records(a.getvalue()) == _records(b.getvalue()) and len(_records(a.getvalue())) > 0


def test_programs_sharded_over_two_ranks_write_the_same_vcf(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m mchap_amd assemble|call-exact ...`: rank r takes its contiguous
    share of the targets / records (mchap_amd.shard.shard_range) before anything touches the GPU, rank 0 gathers the
    formatted lines and writes ONE VCF in target order -- the same records as the single-process run (every unit's
    generator stream depends on the seed alone, so sharding cannot change a result).  Two gloo ranks on the one GPU of the
    test box (RCCL wants a GPU per rank; the exchange is a gather of text lines either way).  Reference: the worker pool
    over blocks of loci in application/baseclass.py:360-388."""
    import subprocess
    import sys

    from mchap_amd import cli

    fa = tmp_path / "simple.fasta"
    fa.write_text("".join(">%s\n%s\n" % (c, "A" * 60) for c in ("CHR1", "CHR2", "CHR3")))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jobs = {
        "assemble": ["--bam"] + [os.path.join(HERE, f) for f in DEEP] + ["--ploidy", "4", "--targets", os.path.join(HERE, "simple.bed"),
                     "--variants", os.path.join(HERE, "simple.vcf"), "--reference", str(fa), "--mcmc-steps", "400", "--mcmc-burn", "100",
                     "--report", "AFP"],
        "call-exact": ["--bam"] + [os.path.join(HERE, f) for f in MIXED] + ["--ploidy", "4", "--haplotypes",
                       os.path.join(HERE, "mock.input.frequencies.vcf"), "--report", "GP"],
    }
    port = 29611
    for program, argv in jobs.items():
        single = _io.StringIO()
        cli.run(["mchap_amd", program] + argv, single)
        env = dict(os.environ, MCHAP_DIST_BACKEND="gloo", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        port += 1
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), "-m", "mchap_amd", program] + argv, capture_output=True, text=True, env=env, cwd=root,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert _records(r.stdout) == _records(single.getvalue()) and len(_records(r.stdout)) >= 2
        head = [ln for ln in r.stdout.splitlines() if ln.startswith("#CHROM")]
        assert len(head) == 1  # one header: only rank 0 writes


def test_call_program_with_device_summaries_writes_the_records_of_the_host_classes(monkeypatch):
    """Round 5: `mchap call` reads GT / GPM / SPM / MCI / AFP / AOP / GP off summaries taken on the device
    (CallingMCMC.fit_batch_summaries); with the summaries formed by the host classes on the downloaded traces instead -- what the
    program did before, and what the reference does (application/call.py:95-160) -- every record line is the same string."""
    from mchap_amd import cli
    from mchap_amd.calling_mcmc import CallingMCMC, CallSummary

    argv = (["mchap_amd", "call", "--bam"] + [os.path.join(HERE, f) for f in DEEP] + ["--ploidy", "4", "--haplotypes",
            os.path.join(HERE, "simple.output.deep.assemble.vcf"), "--mcmc-steps", "500", "--mcmc-burn", "120", "--mcmc-seed", "5",
            "--report", "AFP", "AOP", "GP"])
    a, b = _io.StringIO(), _io.StringIO()
    cli.run(argv, a)

    def by_host(self, reads, read_counts=None, initial=None, haplotypes=None, prior=None, stream_ids=None, burn=0, incongruence_threshold=0.6, max_states=512,
                stream=None):
        return dict(done=[CallSummary.of_trace(t.burn(burn), incongruence_threshold) for t in self.fit_batch(reads, read_counts, initial, haplotypes, prior, stream_ids)])

    monkeypatch.setattr(CallingMCMC, "start_batch_summaries", by_host)
    cli.run(argv, b)
    ra, rb = _records(a.getvalue()), _records(b.getvalue())
    assert len(ra) == len(rb) > 0 and ra == rb
