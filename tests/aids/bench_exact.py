"""Exact caller at BASELINE.json configs[3] (hexaploid, 16 known haplotypes over 10 SNVs, 500 reads, G = 54 264
genotypes per unit): units/s through mchap_exact_posterior_mode_batch (host buffers, so PCIe-inclusive), checked
against the oracle on a few units, with the oracle timed beside it.  Measurement aid, not part of bench.py's contract.

    python tests/bench_exact.py [units]
"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np

from mchap_amd import calling
from mchap_amd.synth import synth_units

U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K, H, M, R = 6, 16, 10, 500
rng = np.random.default_rng(4)
reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, window=(5, 10))
haps = np.unique(rng.integers(0, 2, size=(64, M)).astype(np.int8), axis=0)[:H]
assert len(haps) == H
freq = rng.dirichlet(np.ones(H))
prior = (0.1, freq)
kw = dict(return_support_prob=True, return_posterior_frequencies=True, return_posterior_occurrence=True)
calling.posterior_mode_batch(reads[:8], K, haps, prior=prior, **kw)  # warm up
t = time.perf_counter()
out = calling.posterior_mode_batch(reads, K, haps, prior=prior, **kw)
dt = time.perf_counter() - t
G = calling.count_unique_genotypes(H, K)
print("GPU: %d units x %d genotypes x %d reads in %.3f s -> %.1f units/s, %.3e (genotype, read) terms/s (two passes each)"
      % (U, G, R, dt, U / dt, 2.0 * U * G * R / dt))
try:
    from oracle import binding as orc
except Exception as e:  # noqa: BLE001
    print("oracle not available:", e)
    sys.exit(0)
n = 2
t = time.perf_counter()
for u in range(n):
    a, mllk, mprob, sprob, fr, oc = orc.posterior_mode(reads[u], K, haps, None, prior)
    assert np.array_equal(a, out[0][u]), (a, out[0][u])
    np.testing.assert_allclose([mllk, mprob, sprob], [out[1][u], out[2][u], out[3][u]], rtol=1e-9)
    np.testing.assert_allclose(fr, out[4][u], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(oc, out[5][u], rtol=1e-8, atol=1e-12)
dc = (time.perf_counter() - t) / n
print("oracle (1 thread): %.2f s per unit -> %.2f units/s; GPU/1-thread ratio %.0f; parity ok on %d units" % (dc, 1 / dc, (U / dt) * dc, n))
