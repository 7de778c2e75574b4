"""GPU: targets beyond the library's limits do not end a run.  The reference has no limit on the SNVs of a target; this build
takes 62 SNVs per target and 64 bits of sampled alleles per haplotype (one bit per biallelic SNV, two per tri- / tetra-allelic
one).  `application.assemble` leaves such a target out of the output with a warning on stderr and carries on with the others."""
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


def test_targets_beyond_the_limits_are_left_out_with_a_warning(capsys):
    from mchap_amd import application

    rng = np.random.default_rng(5)
    specs = [("ok1", 100, 6, "AC"), ("toomany", 1000, 70, "AC"), ("toowide", 2000, 40, "ACG"), ("ok2", 3000, 9, "AG"), ("wide_ok", 4000, 44, "AC")]
    targets, variants, matrices = [], [], {}
    for name, start, n, al in specs:
        t, v, m = _target(name, start, n, al, rng)
        targets.append(t)
        variants += v
        matrices[(name, "S1")] = m
    source = application.MatrixSource(["S1"], matrices)
    lines = list(application.assemble(None, variants, {"c1": _NSeq()}, source, ploidy=4, steps=120, burn=60, chains=2, seed=3, targets=targets))
    names = [ln.split("\t")[2] for ln in lines]
    assert names == ["ok1", "ok2", "wide_ok"]   # (44 biallelic SNVs: 176 sub-steps per step, the phased sampler's widest shape)
    err = capsys.readouterr().err
    assert "target toomany" in err and "70 SNVs" in err
    assert "target toowide" in err and "64 bits" in err
    # the records of the targets that ran do not depend on their neighbours
    alone = list(application.assemble(None, variants, {"c1": _NSeq()}, source, ploidy=4, steps=120, burn=60, chains=2, seed=3,
                                      targets=[targets[0], targets[3], targets[4]]))
    assert alone == lines
