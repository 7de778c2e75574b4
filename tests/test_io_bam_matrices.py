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
