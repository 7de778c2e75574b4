"""The block path of application.assemble on the device: DenovoRaggedBatch.from_calls (compact int8 input, descriptors built
with array operations) against the same units given as float64 tensors one by one, and the program's records with the host work
done block-wise (mchap_amd/blockpath.py, application._BlockState) against the per-locus path, line for line."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")


def test_from_calls_equals_the_tensor_batch():
    import torch

    from mchap_amd import DenovoMCMC, encoding
    from mchap_amd.device import DenovoRaggedBatch

    rng = np.random.default_rng(11)
    units, calls, counts = [], [], []
    shapes = [(4, 7, 2, 30), (4, 3, 3, 12), (4, 12, 2, 1), (4, 1, 4, 25), (4, 9, 2, 60), (4, 5, 2, 1)]
    nal_flat, nal_off, r_off, c_off = [], [], [], []
    for i, (K, M, A, R) in enumerate(shapes * 3):
        n_alleles = rng.integers(2, A + 1, size=M) if A > 2 else np.full(M, 2)
        haps = rng.integers(0, n_alleles, size=(K, M))
        c = haps[rng.integers(0, K, size=R)].astype(np.int8)
        c[rng.random(c.shape) < 0.2] = -1
        if R == 1 and i % 2:
            c[:] = -1          # a unit without reads: one all-gap row, no counts
        uc, cn = encoding.unique_counts(c)
        has_counts = not (R == 1 and i % 2)
        units.append(dict(reads=encoding.encode_read_distributions(n_alleles, uc, None, error_rate=0.0024), counts=cn if has_counts else None,
                          n_alleles=n_alleles, ploidy=K, inbreeding=0.1 * (i % 3) if i % 4 else None, stream_id=0, temps=(1.0,)))
        r_off.append(sum(len(x) for x in calls))
        calls.append(uc.reshape(-1))
        c_off.append(sum(len(x) for x in counts) if has_counts else -1)
        if has_counts:
            counts.append(cn)
        nal_off.append(len(nal_flat))
        nal_flat += n_alleles.tolist()
    model = DenovoMCMC(ploidy=4, n_alleles=[2], steps=300, chains=2, random_seed=5)
    a = DenovoRaggedBatch(model, units)
    a.run(100)
    U = len(units)
    b = DenovoRaggedBatch.from_calls(
        model, np.concatenate(calls), np.array(r_off), np.array([len(u["reads"]) for u in units]), np.array([u["reads"].shape[1] for u in units]),
        np.array([u["reads"].shape[2] for u in units]), np.full(U, 4), np.concatenate(counts), np.array(c_off), np.array(nal_flat, dtype=np.int8),
        np.array(nal_off), inbreeding=np.array([np.nan if u["inbreeding"] is None else u["inbreeding"] for u in units]))
    b.run(100)
    torch.cuda.synchronize()
    assert torch.equal(a.d_trace, b.d_trace) and torch.equal(a.d_fixed, b.d_fixed) and torch.equal(a.d_status, b.d_status)
    assert np.array_equal(a.d_llks.cpu().numpy().view(np.uint64), b.d_llks.cpu().numpy().view(np.uint64))
    ra, rb = a.results(), b.results(only=list(range(U)))
    for x, y in zip(ra, rb):
        assert np.array_equal(x["genotypes"], y["genotypes"]) and np.array_equal(x["probabilities"], y["probabilities"]) and x["mci"] == y["mci"]
    arr = b.summary_arrays()
    assert arr["plain"].all() and arr["total"] == 2 * 200
    assert len(arr["words"]) == len(arr["counts"]) == int(arr["n"].sum())
    first = np.cumsum(arr["n"]) - arr["n"]
    for u, x in enumerate(ra):
        assert arr["n"][u] == len(x["genotypes"]) and arr["stats"][u, 1] == x["gpm"] and arr["stats"][u, 0] == x["spm"]
        assert np.array_equal(arr["counts"][first[u]: first[u] + arr["n"][u]] / arr["total"], x["probabilities"])


def _both(tmp_path=None, **kw):
    from mchap_amd import application

    out = []
    for bp in (False, True):
        tm = {}
        out.append((list(application.assemble(block_path=bp, timings=tm, **kw)), tm))
    return out


def test_reference_bams_line_for_line():
    from mchap_amd import io

    from test_gpu_assemble_goldens import SEQ

    ref = {c: SEQ for c in ("CHR1", "CHR2", "CHR3")}
    for names, ploidy, extra in ((("simple.sample1.bam", "simple.sample2.bam", "simple.sample3.bam"), 4, {}),
                                 (("simple.sample1.deep.bam", "simple.sample2.deep.bam", "simple.sample3.deep.bam"), {"SAMPLE1": 4, "SAMPLE2": 2, "SAMPLE3": 6},
                                  dict(report=("AFP", "GP")))):
        bams = {"SAMPLE%d" % (i + 1): os.path.join(HERE, n) for i, n in enumerate(names)}
        (slow, _), (fast, _) = _both(bed_path=os.path.join(HERE, "simple.bed"), variants_vcf_path=os.path.join(HERE, "simple.vcf"),
                                     reference_sequences=ref, sample_bams=bams, ploidy=ploidy, steps=400, burn=200, chains=2, seed=11, **extra)
        assert slow == fast and len(fast) == len(io.read_bed4(os.path.join(HERE, "simple.bed")))


def test_synthetic_job_line_for_line(tmp_path):
    """Two samples of different ploidy and depth (the shallow one's chains wander: units with more distinct genotypes than the
    summary arrays keep go through the listed launch and the per-locus formatter), a tempered sample, several blocks."""
    from mchap_amd import io, synth

    job = synth.synth_assembly_inputs(str(tmp_path), n_loci=50, n_samples=3, reads_per_locus=40, gap=60)
    # the third sample keeps every tenth read only
    bams = dict(zip(("S000", "S001", "S002"), job["bams"]))
    kw = dict(bed_path=job["bed"], variants_vcf_path=job["vcf"], reference_sequences=io.Reference(job["fasta"]), sample_bams=bams,
              ploidy={"S000": 4, "S001": 2, "S002": 4}, inbreeding={"S000": 0.2, "S001": None, "S002": None},
              temperatures={"S000": (1.0,), "S001": (1.0,), "S002": (0.6, 1.0)}, steps=300, burn=150, chains=2, seed=3)
    (slow, _), (fast, tm) = _both(**kw)
    assert slow == fast and len(fast) == 50 and tm["units"] == 150
    (_, _), (blocked, _) = _both(units_per_block=3 * 8, **kw)
    assert blocked == fast
    # a shallow job: 3 reads per locus -- wandering chains, many distinct genotypes per unit
    job2 = synth.synth_assembly_inputs(str(tmp_path / "shallow"), n_loci=30, n_samples=1, reads_per_locus=3, gap=60)
    (slow, _), (fast, _) = _both(bed_path=job2["bed"], variants_vcf_path=job2["vcf"], reference_sequences=io.Reference(job2["fasta"]),
                                 sample_bams={"S000": job2["bams"][0]}, ploidy=4, steps=1500, burn=100, chains=2, seed=3)
    assert slow == fast and len(fast) == 30
