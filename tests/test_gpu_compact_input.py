"""GPU: the compact input of the sampler (int8 allele calls + optional int16 base qualities, turned into the probability
tensor by the prepare pass) gives the same traces as the float64 tensor the host encoders build
(mchap_amd.encoding == reference encoding/integer/transcode.py:16-77, io/bam.py:251-289)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_quals", [False, True])
@pytest.mark.parametrize("n_alleles", [[2, 2, 2, 2, 2, 2], [2, 3, 2, 3, 3, 2, 2]])
def test_calls_input_equals_tensor_input(use_quals, n_alleles):
    import torch
    from mchap_amd import DenovoMCMC, encoding
    from mchap_amd.device import DenovoDeviceBatch

    rng = np.random.default_rng(17)
    U, R, M, K = 6, 70, len(n_alleles), 4
    na = np.array(n_alleles)
    haps = rng.integers(0, na, size=(U, K, M))
    src = haps[np.arange(U)[:, None], rng.integers(0, K, size=(U, R))]
    flip = rng.random(src.shape) < 0.05
    calls = np.where(flip, (src + 1) % na, src).astype(np.int8)
    calls[rng.random(calls.shape) < 0.25] = -1   # gaps
    calls[:, 0, :] = -1                           # an all-gap read
    quals = rng.integers(0, 61, size=calls.shape).astype(np.int16) if use_quals else None
    error_rate = 0.0024
    reads = np.stack([encoding.encode_read_distributions(na, calls[u], None if quals is None else quals[u], error_rate=error_rate)
                      for u in range(U)])
    model = DenovoMCMC(ploidy=K, n_alleles=n_alleles, steps=150, chains=2, random_seed=21)
    a = DenovoDeviceBatch(model, reads)
    b = DenovoDeviceBatch(model, None, calls=calls, quals=quals, error_rate=error_rate)
    a.run()
    b.run()
    torch.cuda.synchronize()
    wa, fa, la, sa = a.traces()
    wb, fb, lb, sb = b.traces()
    assert np.array_equal(sa, sb) and np.array_equal(fa, fb)
    assert np.array_equal(wa, wb)
    assert np.array_equal(la, lb, equal_nan=True)


def test_calls_input_needs_the_prepare_pass(monkeypatch):
    from mchap_amd import DenovoMCMC

    monkeypatch.setenv("MCHAP_HIP_TEST_KERNELS", "1")  # kernel 1 lives in the parity suite's library
    from mchap_amd.device import DenovoDeviceBatch

    calls = np.zeros((1, 8, 3), dtype=np.int8)
    batch = DenovoDeviceBatch(DenovoMCMC(ploidy=2, n_alleles=[2, 2, 2], steps=10, chains=1, random_seed=1, kernel=1), None, calls=calls)
    with pytest.raises(AssertionError):
        batch.run()
