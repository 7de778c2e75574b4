"""GPU: BASELINE.json configs[0] -- `mchap assemble` on docs/example (bi-parental tetraploid population: 20 target loci x 22
samples, 2-23 SNVs per locus, 0-534 read pairs per sample and locus) -- from the pileup fixture
tests/golden/example_biparental.npz (made from the 22 BAM files, targets20.bed and snvs.vcf.gz by
tests/golden/make_example_fixture.py; data only).

  * every (locus x sample) unit's trace against the oracle on the same Philox streams, bit for bit, through the ragged
    batch the program launches (reference assemble/mcmc.py:103-426);
  * the program's VCF records (application.assemble: --report AFP GP, the notebook's settings) against the same program
    with the oracle in place of the kernels, line for line (application/assemble.py:95-252, baseclass.py:220-302);
  * the notebook's own printed records (the real reference, numba's random stream): REF bases at the SNVs, SNV
    positions, NVAR / END, read statistics (DP, RCOUNT, RCALLS: generator-free) exactly, genotype calls where its posterior
    is concentrated."""
import json
import os

import numpy as np
import pytest

from oracle import binding as orc
from tests.helpers import beta_break_table

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEPS, BURN, SEED = 600, 300, 42


def _oracle_sampler(units, settings):
    """The sampler of application.assemble replaced by the oracle + the host classes (GenotypeMultiTrace.posterior,
    mode_genotype_support, replicate_incongruence: mchap_amd/classes.py, pinned to reference vectors elsewhere)."""
    from mchap_amd import GenotypeMultiTrace
    from mchap_amd.classes import sort_haplotypes

    out = []
    for u in units:
        M = u["reads"].shape[1]
        cfg = orc.make_cfg(u["ploidy"], settings["steps"], settings["chains"], u["inbreeding"], u["temps"], llk_cache_threshold=100,
                           rng_kind=orc.RNG_PHILOX, seed=settings["seed"], stream_id=u["stream_id"], break_table=beta_break_table(M, 1.0, 3.0))
        g, l, code = orc.denovo_fit(cfg, u["reads"], list(u["n_alleles"]), u["counts"])
        assert code == 0
        tr = GenotypeMultiTrace._from_sorted(sort_haplotypes(g), l).burn(settings["burn"])
        post = tr.posterior()
        sup = post.mode_genotype_support()
        mg, gp = sup.mode_genotype()
        out.append(dict(genotypes=post.genotypes, probabilities=post.probabilities, spm=float(sup.probabilities.sum()), gpm=float(gp),
                        mode_genotype=mg, mci=int(tr.replicate_incongruence(settings["incongruence_threshold"])), status=0,
                        trace=(sort_haplotypes(g), l)))
    return out


@pytest.fixture(scope="module")
def example():
    from mchap_amd import application

    samples, targets, variants, matrices, contigs = application.load_matrices(os.path.join(GOLDEN, "example_biparental.npz"))
    assert len(samples) == 22 and len(targets) == 20 and contigs == [("chr1", 21898217)]
    return samples, targets, variants, matrices


def _run(example, sampler=None, report=("AFP", "GP"), **kw):
    from mchap_amd import application, io

    samples, targets, variants, matrices = example
    source = application.MatrixSource(samples, matrices)
    ref = {"chr1": _NSeq()}
    return list(application.assemble(None, variants, ref, source, ploidy=4, steps=STEPS, burn=BURN, chains=2, seed=SEED, targets=targets,
                                     report=report, sampler=sampler, **kw))


class _NSeq:
    """a reference sequence known by its length only (io.Reference's behaviour without the FASTA file)"""

    known = False  # (application.assemble: the variants' REF alleles are written at their positions)


    def __getitem__(self, sl):
        return "N" * (sl.stop - sl.start)


def test_program_records_equal_the_oracle_backed_replay(example):
    from mchap_amd.device import DenovoRaggedBatch

    before = DenovoRaggedBatch.n_runs
    got = _run(example)
    assert DenovoRaggedBatch.n_runs - before == 1  # 440 units of 20 different shapes: one ragged launch
    want = _run(example, sampler=_oracle_sampler)
    assert len(got) == 20
    for a, b in zip(got, want):
        assert a == b
    # block-wise processing (bounded memory) gives the same file
    assert _run(example, units_per_block=22 * 3) == got


def test_every_unit_trace_equals_the_oracle(example):
    """The 440 units as application.assemble builds them (de-duplicated rows + counts, phred scores ignored), all in one
    DenovoMCMC.fit_batch call of the default dispatch: traces bit for bit, llks to 1e-10."""
    from mchap_amd import DenovoMCMC, application, io
    from mchap_amd.classes import sort_haplotypes

    samples, targets, variants, matrices = example
    source = application.MatrixSource(samples, matrices)
    by_m = {}
    for contig, start, stop, name in targets:
        locus = io.DenovoLocus(contig, start, stop, name, variants, "N" * (stop - start), sequence_known=False)
        for s in samples:
            sr = source.reads(locus, s)
            if len(sr["dists"]):
                by_m.setdefault(tuple(locus.n_alleles), []).append((sr["dists"], sr["counts"]))
    n_units = 0
    for n_alleles, units in sorted(by_m.items()):
        M = len(n_alleles)
        model = DenovoMCMC(ploidy=4, n_alleles=list(n_alleles), steps=STEPS, chains=2, random_seed=SEED)
        traces = model.fit_batch([u[0] for u in units], [u[1] for u in units], stream_ids=[0] * len(units))
        for (rd, rc), tr in zip(units, traces):
            cfg = orc.make_cfg(4, STEPS, 2, None, (1.0,), llk_cache_threshold=100, rng_kind=orc.RNG_PHILOX, seed=SEED, stream_id=0,
                               break_table=beta_break_table(M, 1.0, 3.0))
            g, l, code = orc.denovo_fit(cfg, rd, list(n_alleles), rc)
            assert code == 0
            assert np.array_equal(tr.genotypes, sort_haplotypes(g)), "M = %d" % M
            np.testing.assert_allclose(tr.llks, l, rtol=1e-10, atol=1e-9, equal_nan=True)
            n_units += 1
    assert n_units >= 400 and len(by_m) >= 8  # (2 to 23 SNVs per locus)


def test_against_the_records_printed_in_the_notebook(example):
    """docs/example/bi-parental.ipynb prints three records of `mchap assemble ... --ploidy 4` (the real reference, 2000
    steps, numba's generator).  What does not depend on the generator must be equal; genotype calls are compared where the
    notebook's posterior mode probability is at least 0.95 (the same multiset of haplotype sequences)."""
    nb = json.load(open(os.path.join(GOLDEN, "example_notebook_records.json")))
    assert sorted(nb) == ["locus001", "locus012", "locus019"]
    lines = {ln.split("\t")[2]: ln.split("\t") for ln in _run(example, report=())}
    checked = agree = 0
    for name, rec in nb.items():
        f = lines[name]
        assert int(f[1]) == rec["pos"] and f[8] == rec["format"]
        info = dict(kv.split("=") for kv in f[7].split(";") if "=" in kv)
        want = dict(kv.split("=") for kv in rec["info"].split(";") if "=" in kv)
        for k in ("DP", "RCOUNT", "END", "NVAR", "SNVPOS", "NS"):
            assert info[k] == want[k], (name, k)
        # REF: the notebook's reference bases at the SNV positions; N elsewhere (no FASTA in the reference's repository)
        for p in (int(x) - 1 for x in want["SNVPOS"].split(",")):
            assert f[3][p] == rec["ref"][p]
        assert set(f[3]) - set("ACGT") <= {"N"} and len(f[3]) == len(rec["ref"])
        snvs = [int(x) - 1 for x in want["SNVPOS"].split(",")]
        alt_nb = {tuple(a[p] for p in snvs): i + 1 for i, a in enumerate(rec["alts"]) if a != "."}
        alt_me = {tuple(a[p] for p in snvs): i + 1 for i, a in enumerate(f[4].split(",")) if a != "."}
        ref_key = tuple(rec["ref"][int(x) - 1] for x in want["SNVPOS"].split(","))
        for col_me, col_nb in zip(f[9:], rec["samples"]):
            a, b = col_me.split(":"), col_nb.split(":")
            assert a[3:6] == b[3:6]  # DP, RCOUNT, RCALLS
            if float(b[8]) >= 0.95 and "." not in b[0]:
                # same multiset of haplotypes: translate allele numbers through the haplotype sequences
                inv_nb = {v: k for k, v in alt_nb.items()}
                inv_me = {v: k for k, v in alt_me.items()}
                g_nb = sorted(ref_key if x == "0" else inv_nb[int(x)] for x in b[0].split("/"))
                g_me = sorted(ref_key if x == "0" else inv_me.get(int(x)) for x in a[0].split("/")) if "." not in a[0] else None
                checked += 1
                agree += int(g_me == g_nb)
    assert checked >= 12 and agree >= 0.85 * checked, (checked, agree)  # (17 confident calls in the three printed records)
