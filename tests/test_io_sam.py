"""SAM text input: the reference reads BAM / SAM through pysam; here the SAM reader must give the
records the BAM reader gives for the reference's own fixture pairs (tests/test_io/data/simple.sample*.{bam,sam})."""
import os

import pytest

from mchap_amd import io

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data")


@pytest.mark.parametrize("sample", ["simple.sample1", "simple.sample2", "simple.sample3"])
def test_sam_records_equal_bam_records(sample):
    bam = io.read_bam(os.path.join(HERE, sample + ".bam"))
    sam = io.read_sam(os.path.join(HERE, sample + ".sam"))
    assert sam[0] == bam[0]  # reference names
    assert sam[1] == bam[1]  # read group -> sample
    assert len(sam[2]) == len(bam[2]) > 0
    for a, b in zip(sam[2], bam[2]):
        assert a == b
    assert io.read_alignments(os.path.join(HERE, sample + ".sam"))[2] == bam[2]
    assert io.read_alignments(os.path.join(HERE, sample + ".bam"))[2] == bam[2]
    assert io.bam_header(os.path.join(HERE, sample + ".sam")) == io.bam_header(os.path.join(HERE, sample + ".bam"))


def test_sample_table_from_sam_paths(tmp_path):
    paths = [os.path.join(HERE, "simple.sample%d.sam" % i) for i in (1, 2, 3)]
    table = io.sample_bam_table(paths)
    assert list(table) == ["SAMPLE1", "SAMPLE2", "SAMPLE3"] and list(table.values()) == paths
    listing = tmp_path / "bams.txt"
    listing.write_text("".join("SAMPLE%d\t%s\n" % (i + 1, p) for i, p in enumerate(paths)))
    assert io.sample_bam_table([str(listing)]) == table
