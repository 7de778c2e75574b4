"""Input encoders against vectors captured from the reference (tests/golden/encoding.npz).  CPU only."""
import os

import numpy as np

from mchap_amd import encoding, synth


def test_as_probabilistic_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "encoding.npz"))
    for fn in (encoding.as_probabilistic, synth.as_probabilistic):
        got = fn(z["calls"], z["n_alleles"], z["p"])
        assert np.array_equal(got, z["probabilistic"], equal_nan=True)
    got = encoding.as_probabilistic(z["calls"], z["n_alleles"], 0.9976)
    assert np.array_equal(got, z["probabilistic_scalar_p"], equal_nan=True)
    # gap at a biallelic position of an A=3 tensor is [nan, nan, 0]
    got = encoding.as_probabilistic(np.array([[0, -1, 2]]), [2, 2, 3], 0.9)
    assert np.array_equal(got, z["survey_example"], equal_nan=True)
    assert np.isnan(got[0, 1, 0]) and np.isnan(got[0, 1, 1]) and got[0, 1, 2] == 0


def test_encode_read_distributions_and_dedup():
    calls = np.array([[0, 1, -1], [0, 1, -1], [1, 1, 0], [0, 1, -1]], dtype=np.int8)
    d = encoding.encode_read_distributions([2, 2, 2], calls)
    assert d.shape == (4, 3, 2)
    assert d[0, 0, 0] == 1 - 0.0024 and d[0, 0, 1] == (1 - (1 - 0.0024)) / 3
    u, c = encoding.unique_counts(d)
    assert len(u) == 2 and c.tolist() == [3, 1]
    q = np.full(calls.shape, 30)
    d2 = encoding.encode_read_distributions([2, 2, 2], calls, q)
    assert d2[0, 0, 0] == (1 - 10 ** (30 / -10)) * (1 - 0.0024)
    assert encoding.encode_read_distributions([2, 2], np.zeros((0, 2), np.int8)).shape == (0, 2, 2)


def test_genotypes_in_vcf_order_matches_the_indexing_functions():
    """The vectorised enumeration (no per-genotype Python loop) against index_as_genotype_alleles / genotype_alleles_as_index,
    themselves pinned to the reference's tables in tests/test_oracle_golden.py."""
    from mchap_amd import calling

    for n_alleles, ploidy in [(2, 2), (3, 4), (5, 3), (7, 6), (16, 2), (1, 4)]:
        G = calling.count_unique_genotypes(n_alleles, ploidy)
        g = calling.genotypes_in_vcf_order(G, ploidy)
        assert g.shape == (G, ploidy)
        assert (np.diff(g, axis=1) >= 0).all() and g.max() == n_alleles - 1
        for i in range(0, G, max(1, G // 50)):
            assert g[i].tolist() == calling.index_as_genotype_alleles(i, ploidy).tolist()
            assert calling.genotype_alleles_as_index(g[i]) == i
