"""Host mirrors of the reference's calling trace / posterior classes (calling/classes.py:127-368) against vectors
captured from the reference (tests/golden/call_mcmc.npz).  CPU only."""
import os

import numpy as np

from mchap_amd.calling_mcmc import GenotypeAllelesMultiTrace
from tests.helpers import assert_same_posterior


def test_trace_summaries_against_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "call_mcmc.npz"))
    for i in range(int(z["n_cases"])):
        p = "c%d_" % i
        H = len(z[p + "haps"])
        tr = GenotypeAllelesMultiTrace(z[p + "trace0_g"], z[p + "trace0_l"], H).burn(10)
        post = tr.posterior()
        assert_same_posterior(post.genotypes[:, :, None], post.probabilities, z[p + "post_g"][:, :, None], z[p + "post_p"])
        m = post.mode(genotype_support=True)
        K = tr.genotypes.shape[2]
        exp = z[p + "mode_support"]
        # the mode genotype is only defined up to ties of equal probabilities
        np.testing.assert_allclose([m[1], m[2]], exp[K:], rtol=1e-12)
        if np.sum(np.isclose(post.probabilities, m[1])) == 1:
            assert np.asarray(m[0]).tolist() == exp[:K].astype(int).tolist()
        fq = tr.posterior_frequencies()
        np.testing.assert_allclose(np.stack(fq), z[p + "post_freqs"], rtol=1e-12)
        arr = post.as_array(H)
        np.testing.assert_allclose(arr, z[p + "as_array"], rtol=1e-12)
        # incongruence: chains whose best two genotypes tie have no defined mode in the reference either
        tie = False
        for ch in tr.split():
            pp = np.sort(ch.posterior().probabilities)[::-1]
            tie = tie or (len(pp) > 1 and pp[0] == pp[1])
        if not tie:
            got = [tr.replicate_incongruence(t) for t in (0.9, 0.6, 0.3)]
            assert got == z[p + "incongruence"].tolist()


def test_relabel_and_mode():
    g = np.array([[[0, 1], [0, 1], [1, 1], [0, 2]]])
    tr = GenotypeAllelesMultiTrace(g, np.zeros((1, 4)), 3)
    post = tr.posterior()
    assert post.genotypes[0].tolist() == [0, 1] and post.probabilities[0] == 0.5
    r = tr.relabel(np.array([2, 0, 1]))
    assert r.genotypes[0, 0].tolist() == [2, 0] and r.n_allele == 3
    assert post.mode()[1] == 0.5
