"""GPU: passes in flight on separate HIP streams (mchap_amd.device.PassesInFlight, what bench.py's timed region
does) give the traces and posterior summaries of the same batches run one after the other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_passes_in_flight_match_sequential():
    import torch

    from mchap_amd import DenovoMCMC
    from mchap_amd.device import DenovoDeviceBatch, PassesInFlight
    from mchap_amd.synth import synth_units

    model = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, steps=200, chains=2, random_seed=7)
    U = 300
    inputs = [synth_units(U, first_unit=1000 * i)[0] for i in range(5)]
    ref = []
    for i, reads in enumerate(inputs):
        b = DenovoDeviceBatch(model, reads, first_stream=1000 * i)
        b.run()
        b.posterior(100)
        torch.cuda.synchronize()
        ref.append((b.d_trace.cpu().numpy().copy(), b.d_llks.cpu().numpy().copy(), b.post["mode_words"].cpu().numpy().copy()))
        del b
    batches = [DenovoDeviceBatch(model, reads, first_stream=1000 * i) for i, reads in enumerate(inputs)]
    flight = PassesInFlight(3)
    # four rounds over the same five batches on three streams: a batch's next pass lands on another stream than its last one and
    # is issued while that one may still be running -- the batch's own event (device._OwnBuffers) orders the two on the device
    # (without it they race on the workspace: seen once in the round-3 suite as a trace that differed)
    for rep in range(4):
        for b in batches[rep:] + batches[:rep]:
            flight.submit(lambda b=b: (b.run(), b.posterior(100)))
    flight.join()
    torch.cuda.synchronize()
    for b, (tr, ll, mw) in zip(batches, ref):
        assert np.array_equal(b.d_trace.cpu().numpy(), tr)
        assert np.array_equal(b.d_llks.cpu().numpy(), ll, equal_nan=True)
        assert np.array_equal(b.post["mode_words"].cpu().numpy(), mw)
