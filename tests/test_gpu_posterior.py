"""GPU parity of the on-device posterior summary (trace_posterior_kernel / trace_incongruence_kernel):
directly against vectors captured from the reference's GenotypeMultiTrace / PosteriorGenotypeDistribution
(tests/golden/trace_posterior.json, assemble/classes.py:265-376), and against the host classes on real sampler traces."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests.helpers import assert_same_posterior

pytestmark = pytest.mark.gpu


def _pack(genotypes, bits=1):
    """int8 [..., K, M] -> uint64 words [..., K]: position 0 most significant (the sampler's trace format)."""
    g = np.asarray(genotypes).astype(np.uint64)
    M = g.shape[-1]
    w = np.zeros(g.shape[:-1], dtype=np.uint64)
    for j in range(M):
        w |= g[..., j] << np.uint64(bits * (M - 1 - j))
    return w


def _unpack(words, M, bits=1):
    w = np.asarray(words, dtype=np.uint64)
    out = np.zeros(w.shape + (M,), dtype=np.int8)
    for j in range(M):
        out[..., j] = ((w >> np.uint64(bits * (M - 1 - j))) & np.uint64((1 << bits) - 1)).astype(np.int8)
    return out


def test_device_posterior_against_reference_goldens(golden_dir):
    """The reference's own outputs for raw traces (sorted states, posterior order and probabilities, mode support,
    SPM / GPM, mode genotype, incongruence codes at four thresholds) from the device kernels: the canonical sort is what
    the sampler writes, so the golden's `sorted` trace is packed and handed to mchap_trace_posterior_batch_device /
    mchap_trace_incongruence_batch_device as they would receive it."""
    import torch
    from mchap_amd import _lib

    with open(os.path.join(golden_dir, "trace_posterior.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 5
    L = _lib.lib()
    seen_codes = set()
    for c in cases:
        srt = np.array(c["sorted"], dtype=np.int8)  # [C, S, K, M], haplotypes in canonical order
        Cn, S, K, M = srt.shape
        burn = int(c["burn"])
        words = _pack(srt)
        # the packed order of a canonically sorted genotype is ascending: the invariant the kernels rely on
        assert (np.diff(words.astype(np.int64), axis=-1) >= 0).all()
        units = np.zeros(1, dtype=_lib.UNIT_DTYPE)
        units[0]["ploidy"], units[0]["trace_off"] = K, 0
        d_units = torch.from_numpy(units.view(np.uint8).reshape(-1)).cuda()
        d_trace = torch.from_numpy(words.view(np.int64).reshape(-1)).cuda()
        ms = 64
        d_words = torch.empty(ms * K, dtype=torch.int64, device="cuda")
        d_counts = torch.empty(ms, dtype=torch.int32, device="cuda")
        d_n = torch.empty(1, dtype=torch.int32, device="cuda")
        d_stats = torch.empty(2, dtype=torch.float64, device="cuda")
        d_mode = torch.empty(1, dtype=torch.int32, device="cuda")
        d_mw = torch.empty(K, dtype=torch.int64, device="cuda")
        d_mc = torch.empty(1, dtype=torch.int32, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        _lib.check(L.mchap_trace_posterior_batch_device(1, p(d_units), S, Cn, burn, p(d_trace), ms, K, p(d_words), p(d_counts),
                                                        p(d_n), p(d_stats), p(d_mode), p(d_mw), p(d_mc), None))
        torch.cuda.synchronize()
        n = int(d_n.item())
        exp_g = np.array(c["post_genotypes"], dtype=np.int8)
        assert n == len(exp_g)
        got_g = _unpack(d_words.cpu().numpy().view(np.uint64).reshape(ms, K)[:n], M)
        total = Cn * (S - burn)
        # states AND their order: probability descending, ties in the reference's (defined) order
        assert_same_posterior(got_g, d_counts.cpu().numpy()[:n] / total, exp_g, c["post_probs"])
        assert d_stats[0].item() == pytest.approx(float(np.sum(c["support_probs"])), rel=1e-14)  # SPM
        assert d_stats[1].item() == c["mode_prob"]                                                 # GPM
        assert np.array_equal(_unpack(d_mw.cpu().numpy().view(np.uint64), M), np.array(c["mode_genotype"], dtype=np.int8))
        assert np.array_equal(got_g[int(d_mode.item())], np.array(c["mode_genotype"], dtype=np.int8))
        assert int(d_mc.item()) / total == c["mode_prob"]
        # incongruence: chains whose two best supports tie have no defined mode support in the reference (unstable
        # argsort, see tests/test_classes.py) and are skipped there as here
        from mchap_amd.classes import GenotypeMultiTrace

        burned = GenotypeMultiTrace._from_sorted(srt, np.zeros((Cn, S))).burn(burn)
        tie = False
        for ch in burned.split():
            top = np.sort(ch.posterior()._support_sums())[::-1]
            tie = tie or (len(top) > 1 and top[0] == top[1])
        if tie:
            continue
        d_mci = torch.empty(1, dtype=torch.int32, device="cuda")
        for thr, expect in c["incongruence"].items():
            _lib.check(L.mchap_trace_incongruence_batch_device(1, p(d_units), S, Cn, burn, p(d_trace), K, C.c_double(float(thr)),
                                                               p(d_mci), None))
            torch.cuda.synchronize()
            assert int(d_mci.item()) == expect, (thr, expect)
            seen_codes.add(expect)
    assert {0, 1, 2} <= seen_codes


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
    # the reference's rule for code 2 (assemble/classes.py:371-375): the union of the qualifying chains' supports holds
    # more haplotypes than the FIRST such chain's support -- {a, b} vs {a, c}: union 3 > 2 -> 2; {a, b, c} vs {a, b}:
    # union 3 == 3 -> 1, although both are tetraploid
    cases = [([a, a, b, b], [a, a, b, b], 0), ([a, a, b, b], [a, b, b, b], 0), ([a, a, b, b], [a, a, a, c], 2),
             ([a, a, b, c], [a, a, b, b], 1), ([a, a, b, b], [c, c, d, e], 2)]
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
    assert 0 in seen and (1 in seen or 2 in seen)


@pytest.mark.parametrize("K,M,steps,chains", [(4, 12, 1400, 2), (2, 16, 900, 3), (8, 6, 700, 2)])
def test_listed_launch_summarises_wandering_chains_like_the_host_classes(K, M, steps, chains):
    """Chains that visit more than 512 distinct genotypes (samples with few or no reads) overflow the batch kernels' table:
    post_n < 0 and mci = -1.  mchap_trace_posterior_listed_device / mchap_trace_incongruence_listed_device summarise those
    units with a table of chains x (steps - burn) states: distinct genotypes in the reference's order with their counts,
    SPM / GPM, mode genotype and the incongruence code equal to the host classes' on the same trace (GenotypeMultiTrace,
    assemble/classes.py:265-376, pinned to reference vectors in tests/test_classes.py); units that did not overflow keep the batch
    call's values."""
    import torch
    from mchap_amd import GenotypeMultiTrace, _lib
    from mchap_amd.classes import sort_haplotypes

    rng = np.random.default_rng(K * 100 + M)
    burn = steps // 4
    U = 4
    traces = []
    for u in range(U):
        if u == 2:  # a settled unit: two genotypes only -- stays with the batch call
            base = rng.integers(0, 2, size=(2, K, M)).astype(np.int8)
            g = base[rng.integers(0, 2, size=(chains, steps))]
        else:       # a random walk: one allele flips per step (as a wandering chain's trace looks), with revisits
            g = np.empty((chains, steps, K, M), dtype=np.int8)
            for c in range(chains):
                cur = rng.integers(0, 2, size=(K, M)).astype(np.int8)
                for s_ in range(steps):
                    if rng.random() < (0.9 if u != 3 else 0.5):
                        cur = cur.copy()
                        cur[rng.integers(0, K), rng.integers(0, M)] ^= 1
                    g[c, s_] = cur
        traces.append(sort_haplotypes(g))
    words = np.concatenate([_pack(t).reshape(-1) for t in traces])
    units = np.zeros(U, dtype=_lib.UNIT_DTYPE)
    for u in range(U):
        units[u]["ploidy"], units[u]["trace_off"] = K, u * chains * steps * K
    L = _lib.lib()
    L.mchap_trace_posterior_max_states.restype = C.c_int
    total = chains * (steps - burn)
    cap = min(total, int(L.mchap_trace_posterior_max_states(K)))
    assert cap == total  # (the LDS holds every state of these runs)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    d_units = torch.from_numpy(units.view(np.uint8).reshape(-1)).cuda()
    d_trace = torch.from_numpy(words.view(np.int64)).cuda()
    ms = 512
    d_words = torch.empty(U * ms * K, dtype=torch.int64, device="cuda")
    d_counts = torch.empty(U * ms, dtype=torch.int32, device="cuda")
    d_n = torch.empty(U, dtype=torch.int32, device="cuda")
    d_stats = torch.empty(U * 2, dtype=torch.float64, device="cuda")
    d_mode = torch.empty(U, dtype=torch.int32, device="cuda")
    d_mw = torch.empty(U * K, dtype=torch.int64, device="cuda")
    d_mc = torch.empty(U, dtype=torch.int32, device="cuda")
    d_mci = torch.empty(U, dtype=torch.int32, device="cuda")
    _lib.check(L.mchap_trace_posterior_batch_device(U, p(d_units), steps, chains, burn, p(d_trace), ms, K, p(d_words), p(d_counts), p(d_n),
                                                    p(d_stats), p(d_mode), p(d_mw), p(d_mc), None))
    _lib.check(L.mchap_trace_incongruence_batch_device(U, p(d_units), steps, chains, burn, p(d_trace), K, C.c_double(0.6), p(d_mci), None))
    torch.cuda.synchronize()
    n0, mci0 = d_n.cpu().numpy().copy(), d_mci.cpu().numpy().copy()
    over = np.flatnonzero((n0 < 0) | (mci0 < 0)).astype(np.int32)
    assert set(over.tolist()) >= {0, 1} and 2 not in over and n0[2] == len(np.unique(_pack(traces[2])[:, burn:].reshape(-1, K), axis=0))
    batch_stats2 = d_stats.cpu().numpy().reshape(U, 2)[2].copy()
    d_list = torch.from_numpy(over).cuda()
    o_words = torch.empty(len(over) * cap * K, dtype=torch.int64, device="cuda")
    o_counts = torch.empty(len(over) * cap, dtype=torch.int32, device="cuda")
    _lib.check(L.mchap_trace_posterior_listed_device(len(over), p(d_list), p(d_units), steps, chains, burn, p(d_trace), cap, K, p(o_words),
                                                     p(o_counts), p(d_n), p(d_stats), p(d_mode), p(d_mw), p(d_mc), None))
    _lib.check(L.mchap_trace_incongruence_listed_device(len(over), p(d_list), p(d_units), steps, chains, burn, p(d_trace),
                                                        min(steps - burn, cap), K, C.c_double(0.6), p(d_mci), None))
    torch.cuda.synchronize()
    n, stats, mci = d_n.cpu().numpy(), d_stats.cpu().numpy().reshape(U, 2), d_mci.cpu().numpy()
    ow = o_words.cpu().numpy().view(np.uint64).reshape(len(over), cap, K)
    oc = o_counts.cpu().numpy().reshape(len(over), cap)
    mw = d_mw.cpu().numpy().view(np.uint64).reshape(U, K)
    assert np.array_equal(stats[2], batch_stats2) and n[2] == n0[2] and mci[2] == mci0[2]
    assert (n[over] > 0).all() and (n[over] > 512).any()
    for i, u in enumerate(over):
        tr = GenotypeMultiTrace._from_sorted(traces[u], np.zeros((chains, steps))).burn(burn)
        post = tr.posterior()
        k = int(n[u])
        assert k == len(post.probabilities)
        assert_same_posterior(_unpack(ow[i, :k], M), oc[i, :k] / total, post.genotypes, post.probabilities)
        sup = post.mode_genotype_support()
        mg, gp = sup.mode_genotype()
        assert abs(stats[u, 0] - sup.probabilities.sum()) < 1e-12 and abs(stats[u, 1] - gp) < 1e-15
        assert np.array_equal(_unpack(mw[u], M), mg)
        assert int(mci[u]) == int(tr.replicate_incongruence(0.6))
