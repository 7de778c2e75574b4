"""extract_read_variants / the allele encoding against the matrices of the reference's own test
(tests/test_io/test_bam.py:43-103, 137-200): simple.sample1.bam at CHR1:5-25 with SNVs at 6, 15 and 22."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from mchap_amd import io

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")

EXPECT_CHARS = np.array([
    ["A", "A", "-"], ["A", "A", "-"], ["C", "G", "-"], ["A", "G", "-"],
    ["A", "A", "A"], ["A", "A", "A"], ["C", "G", "C"], ["A", "G", "T"],
    ["-", "A", "A"], ["-", "A", "A"], ["-", "G", "C"], ["-", "G", "T"],
    ["-", "A", "A"], ["-", "A", "A"], ["-", "G", "C"], ["-", "G", "T"],
    ["-", "-", "A"], ["-", "-", "A"], ["-", "-", "C"], ["-", "-", "T"]], dtype="<U1")
EXPECT_CALLS = np.array([
    [0, 0, -1], [0, 0, -1], [1, 1, -1], [0, 1, -1], [0, 0, 0], [0, 0, 0], [1, 1, 1], [0, 1, 2],
    [-1, 0, 0], [-1, 0, 0], [-1, 1, 1], [-1, 1, 2], [-1, 0, 0], [-1, 0, 0], [-1, 1, 1], [-1, 1, 2],
    [-1, -1, 0], [-1, -1, 0], [-1, -1, 1], [-1, -1, 2]], dtype=np.int8)


def _locus():
    return SimpleNamespace(contig="CHR1", start=5, stop=25, name="CHR1_05_25", sequence="A" * 20,
                           positions=[6, 15, 22], alleles=[("A", "C"), ("A", "G"), ("A", "C", "T")], n_alleles=[2, 2, 3])


@pytest.mark.parametrize("ext", ["bam", "sam"])
def test_read_matrices_of_the_reference_test(ext):
    bam = io.read_alignments(os.path.join(HERE, "simple.sample1." + ext))
    chars, quals = io.extract_read_variants(_locus(), bam, "SAMPLE1")
    np.testing.assert_array_equal(chars, EXPECT_CHARS)
    expect_quals = np.zeros(EXPECT_CHARS.shape, dtype=np.int16)
    expect_quals[EXPECT_CHARS != "-"] = 50
    np.testing.assert_array_equal(quals, expect_quals)
    # the allele calls the application derives from them (encode_sample_reads, application/baseclass.py:140-210)
    from mchap_amd.application import sample_reads

    sr = sample_reads(_locus(), [("SAMPLE1", bam)])
    np.testing.assert_array_equal(sr["calls"], EXPECT_CALLS)
    np.testing.assert_array_equal(sr["depth"], (EXPECT_CHARS != "-").sum(axis=0))
    assert int(sr["counts"].sum()) == len(EXPECT_CHARS)


def test_columnar_reader_and_vectorised_extraction_equal_the_record_reader():
    """The block-wise BAM reader with columnar records (io.BamFile / AlignmentColumns) and the vectorised
    extract_read_variants_columns give the matrices of the per-record reader -- which the test above pins to the reference's
    -- on every locus and sample of the reference's test BAMs (shallow, deep, mixed mapping qualities and flags)."""
    import glob

    bed = io.read_bed4(os.path.join(HERE, "simple.bed"))
    _, variants = io.read_vcf(os.path.join(HERE, "simple.vcf"))
    n = 0
    for path in sorted(glob.glob(os.path.join(HERE, "*.bam"))):
        old = io.read_alignments(path)
        cols = io.BamFile(path).columns()
        assert cols.n == len(old[2])
        for contig, start, stop, name in bed:
            locus = io.DenovoLocus(contig, start, stop, name, variants, "A" * (stop - start))
            for sample in dict.fromkeys(old[1].values()):
                for kw in ({}, dict(min_quality=0, skip_duplicates=False, skip_qcfail=False, skip_supplementary=False), dict(min_quality=61)):
                    a = io.extract_read_variants(locus, old, sample, **kw)
                    b = io.extract_read_variants_columns(locus, cols, sample, **kw)
                    np.testing.assert_array_equal(a[0], b[0])
                    np.testing.assert_array_equal(a[1], b[1])
                    n += 1
    assert n >= 60
    chars, quals = io.extract_read_variants_columns(_locus(), io.BamFile(os.path.join(HERE, "simple.sample1.bam")).columns(), "SAMPLE1")
    np.testing.assert_array_equal(chars, EXPECT_CHARS)


def test_bgzf_blocks_bai_index_and_region_fetch(tmp_path):
    """A synthetic coordinate-sorted BAM written block by block with its .bai (mchap_amd.synth.write_bam): the BGZF walk finds
    every block, the record reader (whole-file gzip inflate) and the block-wise reader see the same records, and a region
    fetch through the index -- only the blocks that can hold overlapping records are inflated -- yields the same matrices as
    the whole file, also for reads whose mates overlap the same SNV (agreeing: qualities added; disagreeing: 'N')."""
    from mchap_amd import synth

    job = synth.synth_assembly_inputs(str(tmp_path), n_loci=40, n_samples=1, reads_per_locus=30, gap=3000)  # 125 kb: 8 index bins
    path = job["bams"][0]
    data = open(path, "rb").read()
    blocks = io.bgzf_blocks(data)
    assert len(blocks) >= 3 and sum(b[1] for b in blocks) == len(data)
    bf = io.BamFile(path, workers=2)
    assert bf.index is not None and bf.refs[0][0] == "chrS" and bf.rg == {"rgS000": "S000"}
    whole = bf.columns()
    old = io.read_alignments(path)
    assert whole.n == len(old[2]) == 40 * 30
    _, variants = io.read_vcf(job["vcf"])
    ref = io.Reference(job["fasta"])
    fetched = 0
    for contig, start, stop, name in io.read_bed4(job["bed"]):
        locus = io.DenovoLocus(contig, start, stop, name, variants, ref.fetch(contig, start, stop))
        a = io.extract_read_variants(locus, old, "S000")
        region = bf.columns(contig, start, stop)
        fetched += region.n
        for cols in (whole, region):
            b = io.extract_read_variants_columns(locus, cols, "S000")
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
        assert len(a[0]) >= 20
    assert fetched < 0.8 * 40 * whole.n  # (a region is a block or two, not the file)
    # overlapping mates: same query name, second record agrees at one SNV and disagrees at another
    locus = io.DenovoLocus("chrS", 100, 160, "t", [dict(chrom="chrS", pos=111, id=".", ref="A", alts=("C",), info={}),
                                                  dict(chrom="chrS", pos=121, id=".", ref="A", alts=("G",), info={})], "A" * 60)
    recs = [dict(qname="p", flag=0, ref=0, pos=105, mapq=60, cigar=[(30, "M")], seq="A" * 5 + "C" + "A" * 9 + "G" + "A" * 14, qual=[30] * 30, rg="g"),
            dict(qname="p", flag=0, ref=0, pos=108, mapq=60, cigar=[(2, "S"), (28, "M")], seq="TT" + "A" * 2 + "C" + "A" * 9 + "A" + "A" * 15, qual=[20] * 30, rg="g")]
    p2 = str(tmp_path / "pair.bam")
    synth.write_bam(p2, [("chrS", 1000)], {"g": "X"}, recs)
    a = io.extract_read_variants(locus, io.read_alignments(p2), "X")
    b = io.extract_read_variants_columns(locus, io.BamFile(p2).columns("chrS", 100, 160), "X")
    assert a[0].tolist() == [["C", "N"]] and a[1].tolist() == [[50, 30]]
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_read_group_behind_a_decoy_tag(tmp_path):
    """A record whose aux bytes hold the literal 'RGZ' inside ANOTHER tag's payload ahead of the real RG:Z tag (a Z string, a B
    array, integer bytes) still belongs to its read group: the reference asks pysam for read.get_tag('RG') (io/bam.py:100-118)."""
    import struct

    from mchap_amd.synth import write_bam

    decoys = [b"XSZ" + b"aRGZs1\0",                                    # a Z string containing "RGZ" + a real group's id
              b"XBBC" + struct.pack("<i", 5) + b"RGZs1",                # a byte array
              b"XIi" + b"RGZ\0",                                        # integer bytes
              b"COZ" + b"xRGZnobody\0"]                                 # names no group at all
    recs = []
    for i, d in enumerate(decoys + [b""]):
        recs.append(dict(qname="q%d" % i, flag=0, ref=0, pos=100 + i, mapq=60, cigar=[(20, "M")], seq="ACGTACGTACGTACGTACGT",
                         qual=[30] * 20, rg="s2", tags_before=d))
    recs.append(dict(qname="q9", flag=0, ref=0, pos=106, mapq=60, cigar=[(20, "M")], seq="ACGTACGTACGTACGTACGT", qual=[30] * 20, rg="s1"))
    path = str(tmp_path / "decoy.bam")
    write_bam(path, [("chrD", 1000)], {"s1": "S1", "s2": "S2"}, recs)
    cols = io.BamFile(path).columns()
    assert [cols.rg_samples[i] for i in cols.rg] == ["S2"] * 5 + ["S1"]
    assert [r["rg"] for r in io.read_bam(path)[2]] == ["s2"] * 5 + ["s1"]
    # region fetches of one file share the last region's columns (one inflate + parse per locus, not per sample)
    bf = io.BamFile(path)
    assert bf.columns("chrD", 100, 130) is bf.columns("chrD", 100, 130)


def test_native_record_walk_equals_the_array_construction(tmp_path):
    """mchap_bam_columns (the library's one-pass walk of the inflated records) against AlignmentColumns.__init__ (array operations,
    itself pinned to the record reader above): every column, the CIGAR table, the read groups (decoy 'RGZ' bytes inside other
    fields included) and the same partition of the records by query name."""
    import glob
    import struct

    from mchap_amd import synth

    job = synth.synth_assembly_inputs(str(tmp_path), n_loci=30, n_samples=1, reads_per_locus=20, gap=50)
    recs = [dict(qname="p", flag=0, ref=0, pos=105, mapq=60, cigar=[(30, "M")], seq="A" * 30, qual=[30] * 30, rg="g"),
            dict(qname="q", flag=1024, ref=0, pos=106, mapq=3, cigar=[(10, "M"), (3, "D"), (5, "I"), (15, "M")], seq="A" * 30, qual=[25] * 30, rg="h"),
            dict(qname="p", flag=16, ref=0, pos=108, mapq=60, cigar=[(2, "S"), (28, "M")], seq="T" * 30, qual=[20] * 30, rg="g",
                 tags_before=b"XAZRGZg\0" + b"XBBc" + struct.pack("<i", 4) + bytes([82, 71, 90, 104]) + b"XCi" + struct.pack("<i", 0x5A4752)),
            dict(qname="r", flag=4, ref=-1, pos=-1, mapq=0, cigar=[], seq="ACGT", qual=[9] * 4, rg="nobody")]
    odd = str(tmp_path / "odd.bam")
    synth.write_bam(odd, [("chrS", 1000)], {"g": "X", "h": "Y"}, recs)
    n_files = 0
    for path in sorted(glob.glob(os.path.join(HERE, "*.bam"))) + job["bams"] + [odd]:
        bf = io.BamFile(path)
        payload = b"".join(io.bgzf_inflate(bf.data, bf.blocks))
        offsets, o = [], bf.header_end
        while o + 4 <= len(payload):
            (block,) = struct.unpack_from("<i", payload, o)
            offsets.append(o)
            o += 4 + block
        for id_field in ("SM", "ID"):
            a = io.AlignmentColumns(bf.refs, bf.rg, payload, offsets, id_field)
            b = io.AlignmentColumns.native(bf.refs, bf.rg, payload, bf.header_end, id_field)
            assert b is not None and a.n == b.n == len(offsets) and a.rg_samples == b.rg_samples
            for name in ("ref_id", "pos", "end", "mapq", "flag", "seq_off", "qual_off", "rg", "seg_first", "c_rec", "c_op", "c_len", "c_ref0",
                         "c_read0", "sort_key"):
                np.testing.assert_array_equal(getattr(a, name), getattr(b, name), err_msg=name)
            assert a.sorted == b.sorted and a.max_span == b.max_span
            # query-name ids: any labels, the same classes
            assert len(np.unique(a.qname)) == len(np.unique(b.qname)) == len(np.unique(np.stack([a.qname, b.qname]), axis=1).T)
        n_files += 1
    assert n_files >= 8


def test_malformed_aux_fields_read_the_same_in_both_walks(tmp_path):
    """Aux data that breaks the SAM specification's layout ahead of (or inside) the RG field: the array construction (io._aux_rg) and
    the library's walk (bam_columns.cpp find_rg) both give 'no read group' -- which reads a sample gets does not depend on whether
    the library loaded."""
    import struct

    from mchap_amd import synth

    ok = b"RGZg\0"
    cases = {
        "unknown B sub-type": b"XBBz" + struct.pack("<i", 2) + b"\1\2" + ok,
        "negative B count": b"XBBc" + struct.pack("<i", -1) + ok,
        "truncated B header": b"XBBc\1",
        "unterminated RG": b"RGZg",
        "unterminated string before RG": b"XAZabc",
        "unknown type": b"XAq\0" + ok,
        "fine": b"XAi" + struct.pack("<i", 7) + ok,
    }
    recs = [dict(qname="r%d" % i, flag=0, ref=0, pos=100 + i, mapq=60, cigar=[(20, "M")], seq="A" * 20, qual=[30] * 20, rg="g", tags=t)
            for i, t in enumerate(cases.values())]
    path = str(tmp_path / "bad_aux.bam")
    synth.write_bam(path, [("chrS", 1000)], {"g": "X"}, recs)
    bf = io.BamFile(path)
    payload = b"".join(io.bgzf_inflate(bf.data, bf.blocks))
    offsets, o = [], bf.header_end
    while o + 4 <= len(payload):
        (block,) = struct.unpack_from("<i", payload, o)
        offsets.append(o)
        o += 4 + block
    a = io.AlignmentColumns(bf.refs, bf.rg, payload, offsets)
    b = io.AlignmentColumns.native(bf.refs, bf.rg, payload, bf.header_end)
    assert b is not None and a.n == b.n == len(cases)
    np.testing.assert_array_equal(a.rg, b.rg)
    assert (a.rg[:-1] < 0).all() and a.rg[-1] >= 0, dict(zip(cases, a.rg.tolist()))
    for (name, t) in cases.items():
        got = io._aux_rg(memoryview(t), 0, len(t))
        assert got == ("g" if name == "fine" else None), name


def test_native_record_walk_on_truncated_and_empty_payloads(tmp_path):
    """A payload cut in the middle of a record (a region's last block) yields the complete records before it; one without any
    record yields empty columns; a record whose own fields overrun its length is refused."""
    import struct

    from mchap_amd import synth

    job = synth.synth_assembly_inputs(str(tmp_path), n_loci=3, n_samples=1, reads_per_locus=10, gap=50)
    bf = io.BamFile(job["bams"][0])
    payload = b"".join(io.bgzf_inflate(bf.data, bf.blocks))
    whole = io.AlignmentColumns.native(bf.refs, bf.rg, payload, bf.header_end)
    assert whole.n == 30
    cut = io.AlignmentColumns.native(bf.refs, bf.rg, payload[:-7], bf.header_end)
    assert cut.n == 29 and np.array_equal(cut.pos, whole.pos[:29]) and np.array_equal(cut.c_len, whole.c_len[: cut.seg_first[-1]])
    none = io.AlignmentColumns.native(bf.refs, bf.rg, payload[: bf.header_end], bf.header_end)
    assert none.n == 0 and none.sorted and len(none.c_rec) == 0 and none.seg_first.tolist() == [0]
    bad = bytearray(payload)
    (first_len,) = struct.unpack_from("<i", payload, bf.header_end)
    struct.pack_into("<i", bad, bf.header_end + 4 + 16, 1 << 20)      # l_seq far beyond the record
    with pytest.raises(ValueError):
        io.AlignmentColumns.native(bf.refs, bf.rg, bytes(bad), bf.header_end)
