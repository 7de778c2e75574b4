"""GPU: the reference's `mchap assemble` golden VCFs (real numba runs: 500 steps, burn 100, 2 chains, seed 11;
tests/test_application_assemble.py:254-440).  Their MCMC draws come from numba's generator, so the comparison is
statistical where the posterior is diffuse (shallow BAMs: same mode genotype, GPM / SPM within the Monte-Carlo error
of the golden's 800 correlated samples) and exact where it is concentrated (deep BAMs: GT, GQ, SQ, GPM, SPM all equal).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")
SEQ = "A" * 60  # tests/test_io/data/simple.fasta: three contigs of 60 A


def _run(bam_files, golden, inbreeding, steps, burn):
    import replay_call_exact as rp
    from mchap_amd import DenovoMCMC

    samples, expect = rp.read_vcf(os.path.join(HERE, golden))
    _, variants = rp.read_vcf(os.path.join(HERE, "simple.vcf"))
    bams = {s: rp.read_bam(os.path.join(HERE, f)) for s, f in zip(samples, bam_files)}
    out = []
    for (contig, start, stop, name), exp in zip(rp.read_bed4(os.path.join(HERE, "simple.bed")), expect):
        assert exp["id"] == name
        locus = rp.DenovoLocus(contig, start, stop, name, variants, SEQ[start:stop])
        gold_seqs = (exp["ref"],) + exp["alts"]
        for s in samples:
            f = dict(zip(exp["format"].split(":"), exp["samples"][s].split(":")))
            calls, uniq, counts = rp.sample_reads(locus, bams[s], s)
            assert str(len(calls)) == f["RCOUNT"] and str(int((calls >= 0).sum())) == f["RCALLS"]
            model = DenovoMCMC(ploidy=4, n_alleles=locus.n_alleles, inbreeding=inbreeding, steps=steps, chains=2, random_seed=11)
            post = model.fit(uniq, read_counts=counts).burn(burn).posterior()
            mode = post.mode_genotype_support() if hasattr(post, "mode_genotype_support") else None
            genotype, gpm = post.genotypes[0], float(post.probabilities[0])
            spm = float(mode.probabilities.sum()) if mode is not None else np.nan
            mine = sorted(locus.format_haplotype(h) for h in genotype)
            gold = sorted(gold_seqs[int(a)] for a in f["GT"].split("/"))
            out.append((name, s, mine, gold, gpm, float(f["GPM"]), spm, float(f["SPM"]), f))
    return out


@pytest.mark.parametrize("golden,inbreeding", [("simple.output.assemble.vcf", 0.0), ("simple.output.assemble.flatprior.vcf", None)])
def test_shallow_assemble_goldens_statistically(golden, inbreeding):
    res = _run(["simple.sample1.bam", "simple.sample2.bam", "simple.sample3.bam"], golden, inbreeding, steps=6000, burn=1000)
    for name, s, mine, gold, gpm, gpm_gold, spm, spm_gold, _ in res:
        assert mine == gold, (name, s)
        assert abs(gpm - gpm_gold) < 0.08 and abs(spm - spm_gold) < 0.08, (name, s, gpm, gpm_gold, spm, spm_gold)


def test_deep_assemble_golden_exactly():
    import replay_call_exact as rp

    res = _run(["simple.sample1.deep.bam", "simple.sample2.deep.bam", "simple.sample3.deep.bam"], "simple.output.deep.assemble.vcf",
               0.0, steps=500, burn=100)
    for name, s, mine, gold, gpm, gpm_gold, spm, spm_gold, f in res:
        assert mine == gold, (name, s)
        assert rp.vcfstr(gpm) == f["GPM"] and rp.vcfstr(spm) == f["SPM"], (name, s, gpm, spm)
        assert str(rp.qual_of_prob(gpm)) == f["GQ"] and str(rp.qual_of_prob(spm)) == f["SQ"]


def test_deep_assemble_golden_whole_records():
    """mchap_amd.application.assemble on the deep BAMs (concentrated posteriors: nothing depends on the generator):
    every record line of the reference's golden VCF, allele order, INFO and MCI included."""
    from mchap_amd import application

    bams = {s: os.path.join(HERE, f) for s, f in zip(["SAMPLE1", "SAMPLE2", "SAMPLE3"],
                                                     ["simple.sample1.deep.bam", "simple.sample2.deep.bam", "simple.sample3.deep.bam"])}
    ref = {c: SEQ for c in ("CHR1", "CHR2", "CHR3")}
    got = list(application.assemble(os.path.join(HERE, "simple.bed"), os.path.join(HERE, "simple.vcf"), ref, bams, ploidy=4,
                                    inbreeding=0.0, steps=500, burn=100, chains=2, seed=11))
    want = [ln.rstrip("\n") for ln in open(os.path.join(HERE, "simple.output.deep.assemble.vcf")) if ln.strip() and not ln.startswith("#")]
    assert got == want


def test_assemble_is_one_sampler_launch_per_vcf():
    """The batched application: every (target x sample) unit of the file in ONE sampler launch, posterior summary and MCI
    from the device kernels (reference application/assemble.py:95-252 loops over samples and loci)."""
    from mchap_amd import application
    from mchap_amd.device import DenovoRaggedBatch

    bams = {s: os.path.join(HERE, f) for s, f in zip(["SAMPLE1", "SAMPLE2", "SAMPLE3"],
                                                     ["simple.sample1.bam", "simple.sample2.bam", "simple.sample3.bam"])}
    ref = {c: SEQ for c in ("CHR1", "CHR2", "CHR3")}
    before = DenovoRaggedBatch.n_runs
    lines = list(application.assemble(os.path.join(HERE, "simple.bed"), os.path.join(HERE, "simple.vcf"), ref, bams, ploidy=4,
                                      inbreeding=0.0, steps=300, burn=100, chains=2, seed=11))
    assert DenovoRaggedBatch.n_runs - before == 1
    assert len(lines) == len(open(os.path.join(HERE, "simple.bed")).read().strip().splitlines())
    # per-sample parameters (the reference's --sample-ploidy / --sample-inbreeding files): a mapping instead of a value;
    # one launch per ploidy present, so that each runs on the fast (single-ploidy) samplers -- and the tetraploid
    # samples' columns are what they were in the all-tetraploid run
    before = DenovoRaggedBatch.n_runs
    mixed = list(application.assemble(os.path.join(HERE, "simple.bed"), os.path.join(HERE, "simple.vcf"), ref, bams,
                                      ploidy={"SAMPLE1": 4, "SAMPLE2": 2, "SAMPLE3": 4}, inbreeding={"SAMPLE1": 0.0, "SAMPLE2": 0.1, "SAMPLE3": 0.0},
                                      steps=200, burn=50, chains=2, seed=11))
    assert DenovoRaggedBatch.n_runs - before == 2
    for ln in mixed:
        cols = ln.split("\t")
        assert len(cols[10].split(":")[0].split("/")) == 2 and len(cols[9].split(":")[0].split("/")) == 4
    # the same job in blocks of one target (three units: two launches per block, blocks pipelined on two side streams,
    # a block's results fetched while the next one runs): the same records
    blocked = list(application.assemble(os.path.join(HERE, "simple.bed"), os.path.join(HERE, "simple.vcf"), ref, bams,
                                        ploidy={"SAMPLE1": 4, "SAMPLE2": 2, "SAMPLE3": 4}, inbreeding={"SAMPLE1": 0.0, "SAMPLE2": 0.1, "SAMPLE3": 0.0},
                                        steps=200, burn=50, chains=2, seed=11, units_per_block=3))
    assert blocked == mixed
