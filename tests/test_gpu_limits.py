"""GPU: wide targets in the program, and targets beyond the library's limits.  The reference has no limit on the SNVs of a target.
This build runs up to 62 SNVs / 64 bits of sampled alleles per haplotype (one bit per biallelic SNV, two per tri- / tetra-allelic
one) on its fast samplers, up to 126 SNVs / 128 bits on the general sampler with 128-bit haplotype words (round 4: such targets used
to be refused) -- and beyond that `application.assemble` writes the target's record with null genotypes and FILTER=LIMIT (declared in
the header), warns on stderr, and the command line exits with status 3; the other targets are not affected."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _NSeq:
    known = False  # (application.assemble: the variants' REF alleles are written at their positions)

    def __getitem__(self, sl):
        return "N" * (sl.stop - sl.start)


def _target(name, start, n_snv, alleles, rng, n_reads=30, ploidy=4):
    """A target of n_snv SNVs 3 bp apart with the given allele letters, and one sample's read characters drawn from `ploidy`
    random haplotypes."""
    pos = [start + 3 * j for j in range(n_snv)]
    variants = [dict(chrom="c1", pos=p + 1, id=".", ref=alleles[0], alts=tuple(alleles[1:]), info={}) for p in pos]
    haps = rng.integers(0, len(alleles), size=(ploidy, n_snv))
    rows = haps[rng.integers(0, ploidy, size=n_reads)]
    chars = np.frombuffer("".join(alleles).encode(), dtype=np.uint8)[rows]
    quals = np.full(chars.shape, 30, dtype=np.int16)
    return ("c1", start, start + 3 * n_snv, name), variants, (chars, quals)


def test_targets_beyond_the_limits_are_written_with_a_filter(capsys):
    from mchap_amd import application

    rng = np.random.default_rng(5)
    specs = [("ok1", 100, 6, "AC"), ("toomany", 1000, 130, "AC"), ("toowide", 2000, 70, "ACG"), ("ok2", 3000, 9, "AG"), ("wide_ok", 4000, 44, "AC"),
             ("wide70", 5000, 70, "AC"), ("wide40x2", 6000, 40, "ACG")]   # (70 biallelic / 40 tri-allelic SNVs: refused until round 4)
    targets, variants, matrices = [], [], {}
    for name, start, n, al in specs:
        t, v, m = _target(name, start, n, al, rng)
        targets.append(t)
        variants += v
        matrices[(name, "S1")] = m
    source = application.MatrixSource(["S1"], matrices)
    lines = list(application.assemble(None, variants, {"c1": _NSeq()}, source, ploidy=4, steps=120, burn=60, chains=2, seed=3, targets=targets))
    names = [ln.split("\t")[2] for ln in lines]
    assert names == ["ok1", "toomany", "toowide", "ok2", "wide_ok", "wide70", "wide40x2"]   # every target has its record, in target order
    by = {ln.split("\t")[2]: ln.split("\t") for ln in lines}
    for nm in ("ok1", "ok2", "wide_ok", "wide70", "wide40x2"):  # (44 biallelic SNVs: the phased sampler's widest shape)
        assert by[nm][6] in ("PASS", "NOA") and "." not in by[nm][9].split(":")[0]
    for nm, n_snv in (("toomany", 130), ("toowide", 70)):
        f = by[nm]
        assert f[4] == "." and f[6] == "LIMIT" and "NVAR=%d;" % n_snv in f[7] and f[7].startswith("AN=0;UAN=0;AC=.;NS=0")
        assert f[8].split(":")[0] == "GT" and f[9].split(":")[0] == "./././." and len(f[9].split(":")) == len(f[8].split(":"))
        assert len(f[3]) == 3 * n_snv  # the reference sequence of the target
    err = capsys.readouterr().err
    assert "target toomany" in err and "130 SNVs" in err
    assert "target toowide" in err and "128 bits" in err
    # the records of the targets that ran do not depend on their neighbours
    alone = list(application.assemble(None, variants, {"c1": _NSeq()}, source, ploidy=4, steps=120, burn=60, chains=2, seed=3,
                                      targets=[targets[0], targets[3], targets[4], targets[5]]))
    assert alone == [lines[0], lines[3], lines[4], lines[5]]
    # ... nor on the host path: the block path formats units of the general sampler from the summary arrays (two words per
    # haplotype: round 5), the per-locus path through results() -- the same record lines
    per_locus = list(application.assemble(None, variants, {"c1": _NSeq()}, source, ploidy=4, steps=120, burn=60, chains=2, seed=3, targets=targets,
                                          block_path=False))
    assert per_locus == lines
    from mchap_amd import vcfheader

    assert any(ln.startswith("##FILTER=<ID=LIMIT,") for ln in vcfheader.header_lines("assemble", "x", ["S1"], [("c1", 9000)]))
    assert any(ln.startswith("##FILTER=<ID=LIMIT,") for ln in vcfheader.header_lines("call-exact", "x", ["S1"], [("c1", 9000)]))  # (round 5)


@pytest.mark.parametrize("block_path", [None, False])
def test_a_deep_wide_target_is_a_limit_record_not_an_abort(capsys, block_path):
    """A target of 70 SNVs runs on the general sampler, which takes 1024 distinct read rows per unit.  One such target with more
    rows used to raise out of the batch constructor and take the whole block with it; now its record says LIMIT and
    the block's other targets -- a wide one that fits among them -- are assembled as if it were not there.  Both host paths."""
    from mchap_amd import application

    rng = np.random.default_rng(11)
    specs = [("ok1", 100, 6, "AC", 30), ("deepwide", 1000, 70, "AC", 1300), ("wide_ok", 5000, 70, "AC", 40), ("ok2", 8000, 8, "AG", 30)]
    targets, variants, matrices = [], [], {}
    for name, start, n, al, nr in specs:
        t, v, m = _target(name, start, n, al, rng, n_reads=nr, ploidy=4)
        if name == "deepwide":  # every read its own row: random characters, not copies of four haplotypes
            chars = np.frombuffer(b"AC", dtype=np.uint8)[rng.integers(0, 2, size=m[0].shape)]
            m = (chars, m[1])
        targets.append(t)
        variants += v
        matrices[(name, "S1")] = m
    source = application.MatrixSource(["S1"], matrices)
    kw = dict(ploidy=4, steps=100, burn=50, chains=2, seed=3, block_path=block_path)
    lines = list(application.assemble(None, variants, {"c1": _NSeq()}, source, targets=targets, **kw))
    by = {ln.split("\t")[2]: ln.split("\t") for ln in lines}
    assert [ln.split("\t")[2] for ln in lines] == ["ok1", "deepwide", "wide_ok", "ok2"]
    assert by["deepwide"][6] == "LIMIT" and by["deepwide"][9].split(":")[0] == "./././."
    for nm in ("ok1", "wide_ok", "ok2"):
        assert by[nm][6] in ("PASS", "NOA") and "." not in by[nm][9].split(":")[0]
    err = capsys.readouterr().err
    assert "target deepwide" in err and "distinct read rows" in err and "1024" in err
    alone = list(application.assemble(None, variants, {"c1": _NSeq()}, source, targets=[targets[0], targets[2], targets[3]], **kw))
    assert alone == [lines[0], lines[2], lines[3]]


def test_call_programs_write_limit_records_instead_of_raising(capsys):
    """Round 5: the exact caller and the call sampler take ploidies up to 15; what they still do not take -- here ploidy 15 over
    120 known haplotypes: 10^19 genotypes, beyond the int64 indices -- used to raise out of the program in the middle of a file.
    Now the record is marked (FILTER=LIMIT, null genotypes: `invalid`), a warning names it, and the other records are called."""
    from types import SimpleNamespace

    from mchap_amd import application, calling

    rng = np.random.default_rng(3)
    M = 6

    def unit(H, K, pos):
        haps = rng.integers(0, 2, size=(H, M)).astype(np.int8)
        reads = rng.dirichlet(np.ones(2), size=(20, M))
        locus = SimpleNamespace(haplotypes=haps, frequencies=np.full(H, 1.0 / H), n_alleles=[2] * M, mask_reference_allele=False)
        return dict(rec=dict(chrom="c1", pos=pos), locus=locus, invalid=None, needs_kernel=True,
                    reads={"S1": dict(dists=reads, counts=np.ones(20, dtype=np.int64))})

    with pytest.raises(NotImplementedError, match="2\\^62"):
        calling.posterior_mode_batch(rng.dirichlet(np.ones(2), size=(1, 20, M)), 15, rng.integers(0, 2, size=(120, M)).astype(np.int8))
    units = [unit(5, 4, 100), unit(120, 15, 200), unit(6, 12, 300)]   # (ploidy 12 over 6 haplotypes: 6 188 genotypes -- runs since round 5)
    # (ploidy is per sample in the programs; three one-record calls keep the test simple)
    out = {}
    for u, K in zip(units, (4, 15, 12)):
        out.update({(u["rec"]["pos"], k[1]): v for k, v in application._run_exact_groups([u], lambda s, K=K: K, lambda s: None, False).items()})
    assert units[0]["invalid"] is None and units[2]["invalid"] is None and units[1]["invalid"] == "LIMIT"
    assert (100, "S1") in out and (300, "S1") in out and (200, "S1") not in out
    assert len(out[(300, "S1")]["alleles"]) == 12
    err = capsys.readouterr().err
    assert "record c1:200 not called" in err and "FILTER=LIMIT" in err
