"""The block-vectorised host path of `mchap assemble` (mchap_amd/blockpath.py) against the per-locus functions it replaces
(io.extract_read_variants_columns, application.encode_reads -- themselves pinned to the reference's matrices in
test_io_bam_matrices.py): same character / quality matrices, calls, depths, distinct rows and counts, cell for cell."""
import glob
import os

import numpy as np
import pytest

from mchap_amd import application, blockpath, io, synth

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")


def _compare(loci, cols, sample, **kw):
    pile = blockpath.extract_block(loci, cols, sample, **kw)
    enc = blockpath.encode_block(loci, pile)
    rows = 0
    for li, locus in enumerate(loci):
        chars, quals = io.extract_read_variants_columns(locus, cols, sample, as_codes=True, **kw)
        c2, q2 = pile.matrices(li)
        np.testing.assert_array_equal(chars, c2)
        np.testing.assert_array_equal(quals, q2)
        sr = application.encode_reads(locus, chars, quals)
        d = enc.per_locus(li)
        np.testing.assert_array_equal(sr["calls"], d["calls"])
        np.testing.assert_array_equal(np.asarray(sr["depth"], dtype=np.int64), d["depth"])
        if len(locus.positions):
            ucalls, counts = application.encoding.unique_counts(np.ascontiguousarray(sr["calls"]))
            np.testing.assert_array_equal(ucalls, d["ucalls"])
            np.testing.assert_array_equal(counts, d["counts"])
            want = np.round(np.mean(sr["depth"]))
            assert enc.dp[li] == want
        else:
            assert np.isnan(enc.dp[li])
        assert enc.rcount[li] == len(sr["calls"]) and enc.rcalls[li] == int((sr["calls"] >= 0).sum())
        rows += len(chars)
    return rows


def test_reference_test_bams():
    bed = io.read_bed4(os.path.join(HERE, "simple.bed"))
    _, variants = io.read_vcf(os.path.join(HERE, "simple.vcf"))
    n = 0
    for path in sorted(glob.glob(os.path.join(HERE, "*.bam"))):
        bf = io.BamFile(path)
        cols = bf.columns()
        loci = [io.DenovoLocus(c, a, b, nm, variants, "A" * (b - a)) for c, a, b, nm in bed]
        loci.append(io.DenovoLocus(bed[0][0], bed[0][1], bed[0][2], "nosnv", [], "A" * (bed[0][2] - bed[0][1])))   # a target without SNVs
        loci.append(io.DenovoLocus("nowhere", 5, 50, "nocontig", [], "A" * 45))
        for sample in dict.fromkeys(bf.rg.values()):
            for kw in ({}, dict(min_quality=0, skip_duplicates=False, skip_qcfail=False, skip_supplementary=False), dict(min_quality=61)):
                n += _compare(loci, cols, sample, **kw)
    assert n > 100


def test_synthetic_job_with_overlapping_mates(tmp_path):
    job = synth.synth_assembly_inputs(str(tmp_path), n_loci=60, n_samples=2, reads_per_locus=25, gap=40)
    _, variants = io.read_vcf(job["vcf"])
    ref = io.Reference(job["fasta"])
    by = application._variants_by_contig(variants)
    loci = [io.DenovoLocus(c, a, b, nm, application._variants_within(by, c, a, b), ref.fetch(c, a, b)) for c, a, b, nm in io.read_bed4(job["bed"])]
    for path, sample in zip(job["bams"], ("S000", "S001")):
        assert _compare(loci, io.BamFile(path).columns(), sample) > 60 * 20
        assert _compare(loci, io.BamFile(path).columns(), "someone else") == 0
    # mates that overlap the same SNVs (agreeing: qualities added; disagreeing: 'N'; a third record of the name: applied in order),
    # tri-allelic SNVs, soft clips and deletions, two loci sharing reads
    var = [dict(chrom="chrS", pos=111, id=".", ref="A", alts=("C",), info={}), dict(chrom="chrS", pos=121, id=".", ref="A", alts=("G", "T"), info={}),
           dict(chrom="chrS", pos=131, id=".", ref="A", alts=("T",), info={})]
    loci = [io.DenovoLocus("chrS", 100, 160, "t", var, "A" * 60), io.DenovoLocus("chrS", 115, 140, "u", var, "A" * 25),
            io.DenovoLocus("chrS", 400, 460, "empty", [], "A" * 60)]
    recs = [dict(qname="p", flag=0, ref=0, pos=105, mapq=60, cigar=[(30, "M")], seq="A" * 5 + "C" + "A" * 9 + "G" + "A" * 14, qual=[30] * 30, rg="g"),
            dict(qname="q", flag=0, ref=0, pos=106, mapq=60, cigar=[(10, "M"), (3, "D"), (20, "M")], seq="A" * 4 + "C" + "A" * 25, qual=[25] * 30, rg="g"),
            dict(qname="p", flag=0, ref=0, pos=108, mapq=60, cigar=[(2, "S"), (28, "M")], seq="TT" + "A" * 2 + "C" + "A" * 9 + "A" + "A" * 15, qual=[20] * 30, rg="g"),
            dict(qname="p", flag=0, ref=0, pos=109, mapq=60, cigar=[(30, "M")], seq="A" + "C" + "A" * 9 + "T" + "A" * 18, qual=[7] * 30, rg="g"),
            dict(qname="z", flag=0, ref=0, pos=118, mapq=60, cigar=[(30, "M")], seq="AAT" + "A" * 9 + "T" + "A" * 17, qual=[9] * 30, rg="h")]
    p2 = str(tmp_path / "pair.bam")
    synth.write_bam(p2, [("chrS", 1000)], {"g": "X", "h": "Y"}, recs)
    cols = io.BamFile(p2).columns()
    assert _compare(loci, cols, "X") == 4 and _compare(loci, cols, "Y") == 2
    pile = blockpath.extract_block(loci, cols, "X")
    assert "N" in pile.matrices(0)[0].tobytes().decode()


def test_unsorted_positions_are_left_to_the_per_locus_path():
    var = [dict(chrom="c", pos=20, id=".", ref="A", alts=("C",), info={}), dict(chrom="c", pos=10, id=".", ref="A", alts=("G",), info={})]
    with pytest.raises(blockpath.BlockPathUnavailable):
        blockpath.locus_tables([io.DenovoLocus("c", 0, 60, "t", var, "A" * 60)])


def test_unpack_words_equals_unpack_trace():
    from mchap_amd.assemble import unpack_trace

    rng = np.random.default_rng(5)
    M = np.array([5, 1, 9, 3])
    A = np.array([2, 4, 3, 2])
    bits = np.array([1, 2, 2, 1])
    fixed = [np.where(rng.random(m) < 0.4, rng.integers(0, a, size=m), -1).astype(np.int8) for m, a in zip(M, A)]
    fixed_off = np.cumsum(M) - M
    unit = rng.integers(0, 4, size=50)
    words = np.array([rng.integers(0, 1 << int(bits[u] * (fixed[u] < 0).sum())) for u in unit], dtype=np.uint64)
    flat, cell_of = blockpath.unpack_words(words, unit, np.concatenate(fixed), fixed_off, M, bits)
    for i, (w, u) in enumerate(zip(words, unit)):
        np.testing.assert_array_equal(flat[cell_of[i]: cell_of[i] + M[u]], unpack_trace(np.array([w], dtype=np.uint64), fixed[u], int(A[u]))[0])


# ---- the whole block path of application.assemble against the per-locus path, with a stand-in for the device batch ----
def _fake_summary(ucalls, K, n_alleles, total):
    """A deterministic, shape-respecting stand-in for a unit's posterior summary, a function of the unit's distinct call rows:
    (words [n, K], counts [n], fixed [M], spm, gpm, mode_words [K], mci, plain)."""
    import zlib

    ucalls = np.asarray(ucalls, dtype=np.int8)
    M = ucalls.shape[1]
    rng = np.random.default_rng(zlib.crc32(ucalls.tobytes()) + 7 * K + M)
    A = int(max(n_alleles))
    bits = 1 if A <= 2 else 2 if A <= 4 else 3
    fixed = np.where(rng.random(M) < 0.3, rng.integers(0, np.asarray(n_alleles)), -1).astype(np.int8)
    het = np.flatnonzero(fixed < 0)
    n = int(rng.integers(1, 6))
    haps = [np.where(fixed < 0, 0, fixed)]                      # the reference allele at every sampled position ...
    for _ in range(3):
        h = haps[0].copy()
        h[het] = rng.integers(0, np.asarray(n_alleles)[het]) if len(het) else h[het]
        haps.append(h)
    if rng.random() < 0.3:
        haps = haps[1:]                                           # ... or a unit that never holds it
    g = np.array([[haps[i] for i in sorted(rng.integers(0, len(haps), size=K))] for _ in range(n)], dtype=np.int8)
    sh = (bits * (len(het) - 1 - np.arange(len(het)))).astype(np.uint64)
    words = (g[:, :, het].astype(np.uint64) << sh).sum(axis=2).astype(np.uint64) if len(het) else np.zeros((n, K), dtype=np.uint64)
    # distinct genotypes only (a posterior lists each once)
    _, keep = np.unique(words, axis=0, return_index=True)
    words = words[np.sort(keep)]
    n = len(words)
    cuts = np.sort(rng.integers(1, total, size=n - 1)) if n > 1 else np.zeros(0, dtype=np.int64)
    counts = np.sort(np.diff(np.r_[0, cuts, total]))[::-1].astype(np.int32)
    gpm = counts[0] / total
    spm = min(1.0, gpm + float(rng.random()) * (1 - gpm))
    return words, counts, fixed, spm, gpm, words[0], int(rng.integers(0, 3)), bool(rng.random() < 0.9)


class _FakeBatch:
    """What application._BlockState needs of device.DenovoRaggedBatch.from_calls, computed by _fake_summary."""

    wph = 1
    MS = 8

    def __init__(self, model, calls, reads_off, n_reads, n_pos, max_allele, ploidy, counts, counts_off, n_alleles, nalleles_off,
                 inbreeding=None, error_rate=0.0024):
        from mchap_amd.assemble import unpack_trace

        U = len(n_reads)
        self.total = model.chains * (model.steps - 0)
        self.model = model
        self.units_host = dict(fixed_off=np.cumsum(n_pos) - n_pos)
        self.K = int(ploidy.max())
        self.args = (calls, reads_off, n_reads, n_pos, max_allele, n_alleles, nalleles_off, U)
        assert (counts_off[n_reads > 1] >= 0).all() and len(inbreeding) == U

    def run(self, burn, incongruence_threshold=0.6):
        self.total = self.model.chains * (self.model.steps - burn)

    def _all(self):
        from mchap_amd.assemble import unpack_trace

        calls, reads_off, n_reads, n_pos, max_allele, n_alleles, nalleles_off, U = self.args
        out = []
        for u in range(U):
            R, M = int(n_reads[u]), int(n_pos[u])
            uc = calls[reads_off[u]: reads_off[u] + R * M].reshape(R, M)
            out.append(_fake_summary(uc, self.K, n_alleles[nalleles_off[u]: nalleles_off[u] + M].tolist(), self.total) + (int(max_allele[u]),))
        return out

    def summary_arrays(self):
        res = self._all()
        U, K, ms = len(res), self.K, self.MS
        a = dict(n=np.zeros(U, dtype=np.int32), stats=np.zeros((U, 2)), mode_words=np.zeros((U, K), dtype=np.uint64), mci=np.zeros(U, dtype=np.int32),
                 status=np.zeros(U, dtype=np.int32), total=self.total)
        for u, (w, c, fx, spm, gpm, mw, mci, plain, A) in enumerate(res):
            a["n"][u] = len(w) if plain else -1
            a["stats"][u], a["mode_words"][u], a["mci"][u] = (spm, gpm), mw, mci
        a["fixed"] = np.concatenate([r[2] for r in res])
        a["plain"] = a["n"] >= 0
        a["words"] = np.concatenate([r[0] for r in res if r[7]] + [np.zeros((0, K), dtype=np.uint64)])   # (the plain units' rows only)
        a["counts"] = np.concatenate([r[1] for r in res if r[7]] + [np.zeros(0, dtype=np.int32)])
        return a

    def results(self, raise_on_limit=True, only=None):
        res = self._all()
        return [_as_dict(*res[u], total=self.total) for u in (range(len(res)) if only is None else only)]


def _as_dict(w, c, fx, spm, gpm, mw, mci, plain, A, total):
    from mchap_amd.assemble import unpack_trace

    return dict(genotypes=unpack_trace(w, fx, A), probabilities=c / total, spm=float(spm), gpm=float(gpm), mode_genotype=unpack_trace(mw, fx, A),
                mci=mci, status=0)


def _fake_sampler(units, settings):
    out = []
    total = settings["chains"] * (settings["steps"] - settings["burn"])
    for u in units:
        d = np.asarray(u["reads"])
        calls = np.where(np.isnan(d).any(axis=2), -1, np.nan_to_num(d).argmax(axis=2)).astype(np.int8)
        out.append(_as_dict(*_fake_summary(calls, u["ploidy"], list(u["n_alleles"]), total), int(d.shape[2]), total=total))
    return out


@pytest.mark.parametrize("report", [(), ("AFP", "GP", "SNVDP")])
def test_assemble_lines_equal_the_per_locus_path(tmp_path, report, monkeypatch, capsys):
    job = synth.synth_assembly_inputs(str(tmp_path), n_loci=40, n_samples=3, reads_per_locus=12, n_snvs=6, gap=60)
    _, variants = io.read_vcf(job["vcf"])
    for r in variants[::7]:
        r["alts"] = r["alts"] + tuple(x for x in "ACGT" if x not in (r["ref"],) + r["alts"])[:1]   # some tri-allelic SNVs
    targets = io.read_bed4(job["bed"])
    last = targets[-1]
    targets += [("chrS", last[2] + 2, last[2] + 30, "no_snvs"), ("chrS", 3, 40, "no_reads_no_snvs"),
                ("chrS", targets[0][1], targets[1][2], "two_windows")]          # 12 SNVs, reads of two windows
    variants.append(dict(chrom="chrS", pos=20, id=".", ref="A", alts=("C",), info={}))   # an SNV nobody covers (no reads: one gap row)
    targets.append(("chrS", 15, 30, "snv_without_reads"))
    monkeypatch.setattr(application, "MAX_SNVS_PER_LOCUS", 11)                    # "two_windows" is beyond the limit
    bams = dict(zip(("S000", "S001", "S002"), job["bams"]))
    ref = io.Reference(job["fasta"])
    kw = dict(ploidy={"S000": 4, "S001": 2, "S002": 4}, inbreeding={"S000": 0.1, "S001": None, "S002": 0.0}, steps=60, burn=20, chains=2,
              temperatures={"S000": (1.0,), "S001": (1.0,), "S002": (0.5, 1.0)}, report=report, targets=targets)
    slow = list(application.assemble(None, variants, ref, application.ReadSource(bams), sampler=_fake_sampler, **kw))
    err_slow = capsys.readouterr().err
    tm = {}
    fast = list(application.assemble(None, variants, ref, application.ReadSource(bams), block_path=True, _batch_factory=_FakeBatch, timings=tm, **kw))
    err_fast = capsys.readouterr().err
    assert len(fast) == len(slow) == len(targets)
    for a, b in zip(slow, fast):
        assert a == b
    assert err_slow == err_fast and "two_windows" in err_fast and tm["limit_records"] == 1
    assert sum("LIMIT" in x for x in fast) == 1 and any("\tNOA\t" in x or "REFMASKED" in x for x in fast)
    # several blocks: the same lines
    again = list(application.assemble(None, variants, ref, application.ReadSource(bams, workers=3), block_path=True, _batch_factory=_FakeBatch,
                                      units_per_block=3 * 7, **kw))   # (workers: the files are inflated and walked side by side)
    assert again == fast


def test_matrix_source_goes_the_block_path_too(tmp_path):
    """Pileups handed over as matrices (application.MatrixSource: the docs/example fixture): same lines either way."""
    job = synth.synth_assembly_inputs(str(tmp_path), n_loci=12, n_samples=2, reads_per_locus=15, n_snvs=5, gap=60)
    _, variants = io.read_vcf(job["vcf"])
    targets = io.read_bed4(job["bed"])
    ref = io.Reference(job["fasta"])
    by = application._variants_by_contig(variants)
    matrices = {}
    for c, a, b, nm in targets:
        locus = io.DenovoLocus(c, a, b, nm, application._variants_within(by, c, a, b), ref.fetch(c, a, b))
        for path, sample in zip(job["bams"], ("S000", "S001")):
            ch, q = io.extract_read_variants_columns(locus, io.BamFile(path).columns(), sample)
            matrices[(nm, sample)] = (ch, q)   # ('U1' characters, as extract_read_variants returns them)
    kw = dict(ploidy=4, steps=60, burn=20, chains=2, targets=targets)
    src = application.MatrixSource(["S000", "S001"], matrices)
    slow = list(application.assemble(None, variants, ref, src, sampler=_fake_sampler, **kw))
    fast = list(application.assemble(None, variants, ref, src, block_path=True, _batch_factory=_FakeBatch, **kw))
    from_bam = list(application.assemble(None, variants, ref, application.ReadSource(dict(zip(("S000", "S001"), job["bams"]))), block_path=True,
                                         _batch_factory=_FakeBatch, **kw))
    assert slow == fast == from_bam and len(fast) == 12


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_alignments(tmp_path, seed):
    """Random coordinate-sorted files -- two contigs, three read groups of two samples, CIGARs with soft clips, insertions, deletions,
    skips and = / X runs, mates that overlap, duplicates / QC failures / supplementary records, low mapping qualities, reads
    without a read group, unplaced reads -- and random targets (overlapping each other, tri-allelic SNVs, some without SNVs):
    the block extraction and encoding against the per-locus functions."""
    rng = np.random.default_rng(seed)
    contigs = [("c1", 3000), ("c2", 1500)]
    groups = {"g1": "A", "g2": "B", "g3": "A"}
    recs = []
    for ci, (cn, ln) in enumerate(contigs):
        for q in range(220):
            name = "q%d_%d" % (ci, rng.integers(0, 150))     # names repeat: mates
            pos = int(rng.integers(0, ln - 150))
            ops, rl = [], 0
            if rng.random() < 0.2:
                ops.append((int(rng.integers(1, 6)), "S"))
            for _ in range(int(rng.integers(1, 5))):
                ops.append((int(rng.integers(5, 40)), str(rng.choice(["M", "M", "M", "=", "X"]))))
                k = rng.random()
                if k < 0.2:
                    ops.append((int(rng.integers(1, 5)), "I"))
                elif k < 0.4:
                    ops.append((int(rng.integers(1, 8)), "D"))
                elif k < 0.45:
                    ops.append((int(rng.integers(5, 30)), "N"))
            while ops and ops[-1][1] in "IDN":
                ops.pop()
            if rng.random() < 0.2:
                ops.append((int(rng.integers(1, 6)), "S"))
            rl = sum(l for l, o in ops if o in "MIS=X")
            flag = int(rng.choice([0, 0, 0, 16, 1024, 512, 2048, 4]))
            recs.append(dict(qname=name, flag=flag, ref=(-1 if flag == 4 else ci), pos=(-1 if flag == 4 else pos), mapq=int(rng.choice([60, 60, 30, 5])),
                             cigar=([] if flag == 4 else ops), seq="".join(rng.choice(list("ACGT"), size=rl)), qual=rng.integers(2, 41, size=rl).tolist(),
                             rg=str(rng.choice(["g1", "g2", "g3", "unknown"]))))
    recs.sort(key=lambda r: (r["ref"] if r["ref"] >= 0 else 99, r["pos"]))
    path = str(tmp_path / "r.bam")
    synth.write_bam(path, contigs, groups, recs)
    cols = io.BamFile(path).columns()
    assert cols.sorted and cols.n == len(recs)
    variants = []
    for cn, ln in contigs:
        for p in np.sort(rng.choice(np.arange(10, ln - 10), size=ln // 25, replace=False)):
            ref = str(rng.choice(list("ACGT")))
            alts = tuple(str(x) for x in rng.permutation([b for b in "ACGT" if b != ref])[: int(rng.integers(1, 4))])
            variants.append(dict(chrom=cn, pos=int(p) + 1, id=".", ref=ref, alts=alts, info={}))
    by = application._variants_by_contig(variants)
    loci = []
    for t in range(60):
        cn, ln = contigs[int(rng.integers(0, 2))]
        a = int(rng.integers(0, ln - 200))
        b = a + int(rng.integers(5, 200))
        loci.append(io.DenovoLocus(cn, a, b, "t%d" % t, application._variants_within(by, cn, a, b), "N" * (b - a)))
    total = 0
    for sample in ("A", "B", "nobody"):
        for kw in ({}, dict(min_quality=0, skip_duplicates=False, skip_qcfail=False, skip_supplementary=False)):
            total += _compare(loci, cols, sample, **kw)
    assert total > 500
