"""GPU parity: on-device posterior summary vs the host classes (which mirror the reference's
GenotypeMultiTrace / PosteriorGenotypeDistribution and are themselves pinned by tests/test_classes.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_posterior_matches_host_classes():
    import torch
    from mchap_amd import DenovoMCMC, GenotypeMultiTrace
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    for (K, M, R, steps, burn, chains) in [(4, 8, 200, 300, 100, 2), (2, 5, 12, 200, 50, 3), (6, 5, 30, 120, 20, 2)]:
        reads, _, _ = synth_units(12, ploidy=K, n_pos=M, n_reads=R, first_unit=3, window=(2, M))
        model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=steps, chains=chains, random_seed=5)
        batch = DenovoDeviceBatch(model, reads)
        batch.run()
        batch.posterior(burn, max_states=64)
        torch.cuda.synchronize()
        words, fixed, llks, status = batch.traces()
        post = batch.posterior_host()
        for u in range(len(reads)):
            g = batch.genotypes(u, words, fixed)
            trace = GenotypeMultiTrace._from_sorted(g, llks[u]).burn(burn)
            ref = trace.posterior()
            n = len(ref.genotypes)
            assert post["n"][u] == n
            k = min(n, 64)
            from mchap_amd.assemble import unpack_trace

            got = unpack_trace(post["words"][u][:k], fixed[u], 2)
            assert np.array_equal(got, ref.genotypes[:k])
            np.testing.assert_allclose(post["counts"][u][:k] / (chains * (steps - burn)), ref.probabilities[:k], rtol=1e-15)
            sup = ref.mode_genotype_support()
            mg, mp = sup.mode_genotype()
            assert post["stats"][u][0] == pytest.approx(sup.probabilities.sum(), rel=1e-14)
            assert post["stats"][u][1] == pytest.approx(mp, rel=1e-15)
            assert np.array_equal(got[post["mode"][u]], mg)


def test_device_incongruence_matches_host_classes():
    """MCI on the device vs GenotypeMultiTrace.replicate_incongruence (reference assemble/classes.py:341-376), on real
    traces (shallow reads: chains disagree now and then) and on constructed ones covering the codes 0, 1 and 2."""
    import ctypes as C
    import torch
    from mchap_amd import DenovoMCMC, GenotypeMultiTrace, _lib
    from mchap_amd.device import DenovoDeviceBatch
    from mchap_amd.synth import synth_units

    seen = set()
    for (K, M, R, steps, burn, chains, thr) in [(4, 6, 10, 120, 20, 3, 0.6), (2, 5, 6, 100, 10, 4, 0.3), (4, 8, 200, 200, 50, 2, 0.6)]:
        reads, _, _ = synth_units(40, ploidy=K, n_pos=M, n_reads=R, first_unit=11, window=(2, M), qual=(3, 15))
        model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=steps, chains=chains, random_seed=9)
        batch = DenovoDeviceBatch(model, reads)
        batch.run()
        mci = batch.incongruence(burn, thr).cpu().numpy()
        words, fixed, llks, _ = batch.traces()
        for u in range(len(reads)):
            trace = GenotypeMultiTrace._from_sorted(batch.genotypes(u, words, fixed), llks[u]).burn(burn)
            expect = trace.replicate_incongruence(thr)
            assert mci[u] == expect, (K, M, u)
            seen.add(int(expect))
    # constructed traces: chain 0 sits on {a, a, b, b}, chain 1 on {a, a, b, b} / {a, b, b, b} / {c, c, d, e}
    L = _lib.lib()
    K, S, Cn = 4, 10, 2
    a, b, c, d, e = 1, 2, 4, 7, 9
    cases = [([a, a, b, b], [a, a, b, b], 0), ([a, a, b, b], [a, b, b, b], 0), ([a, a, b, b], [a, a, a, c], 1), ([a, a, b, b], [c, c, d, e], 2)]
    units = np.zeros(len(cases), dtype=_lib.UNIT_DTYPE)
    trace = np.zeros((len(cases), Cn, S, K), dtype=np.uint64)
    for i, (g0, g1, _) in enumerate(cases):
        trace[i, 0], trace[i, 1] = sorted(g0), sorted(g1)
        units[i]["trace_off"], units[i]["ploidy"] = i * Cn * S * K, K
    d_units = torch.from_numpy(units.view(np.uint8).reshape(-1)).cuda()
    d_trace = torch.from_numpy(trace.view(np.int64).reshape(-1)).cuda()
    d_mci = torch.empty(len(cases), dtype=torch.int32, device="cuda")
    _lib.check(L.mchap_trace_incongruence_batch_device(
        len(cases), C.c_void_p(d_units.data_ptr()), S, Cn, 0, C.c_void_p(d_trace.data_ptr()), K, C.c_double(0.6),
        C.c_void_p(d_mci.data_ptr()), None))
    torch.cuda.synchronize()
    assert d_mci.cpu().tolist() == [x[2] for x in cases]
    assert 0 in seen
