"""GPU: units beyond what the fast samplers take -- more than 62 SNVs, more than 64 bits of sampled alleles per haplotype, or a
ploidy above 8.  The reference has no such limits (assemble/mcmc.py:24-40); until round 4 this build refused these shapes.  They now
run on the lanes-over-chains sampler instantiated with 128-bit haplotype words and sixteen-nibble label packs
(denovo_simt_kernel<0, u128>: up to 126 SNVs / 128 bits / ploidy 15), and their traces hold two uint64 words per haplotype.  Parity as everywhere: the oracle on the same Philox streams, step for step --
int8 genotypes bit-exact, llks to 1e-10."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    # name: (units, ploidy, n_pos, n_reads, n_alleles, inbreeding, temperatures, synth kwargs)
    "tetraploid-80-snvs": (3, 4, 80, 60, 2, None, (1.0,), dict(window=(20, 80))),
    "diploid-120-snvs": (3, 2, 120, 50, 2, None, (1.0,), dict(window=(30, 120))),
    "triploid-50-triallelic": (3, 3, 50, 40, 3, 0.1, (1.0,), dict(window=(10, 50))),       # two bits per allele: 100 bits
    "tetraploid-70-snvs-ladder": (2, 4, 70, 40, 2, None, (0.1, 1.0), dict(window=(20, 70))),
    "octoploid-16-tetraallelic": (2, 8, 16, 40, 4, 0.0, (1.0,), dict(window=(6, 16))),      # 32 bits per haplotype: a 64-bit sampler, wide cache keys
    "tetraploid-63-snvs": (2, 4, 63, 40, 2, None, (1.0,), dict(window=(20, 63))),           # 63 bits, but more than 62 positions
    # ploidies above 8 (packs of sixteen nibbles for labels and doses): the same general kernel
    "decaploid-6-snvs": (3, 10, 6, 60, 2, None, (1.0,), dict(window=(3, 6))),
    "dodecaploid-5-triallelic-inbred": (2, 12, 5, 50, 3, 0.2, (1.0,), dict(window=(2, 5))),
    "ploidy-15-null-prior-ladder": (2, 15, 4, 40, 2, 0.0, (0.2, 1.0), dict(window=(2, 4))),
}


@pytest.mark.parametrize("case", list(CASES))
def test_wide_units_against_the_oracle(case):
    from oracle import binding as orc

    from mchap_amd import DenovoMCMC
    from mchap_amd.assemble import break_table
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    U, K, M, R, A, F, temps, skw = CASES[case]
    reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=900, **skw)
    steps = 25
    model = DenovoMCMC(ploidy=K, n_alleles=[A] * M, steps=steps, chains=2, inbreeding=F, temperatures=temps, random_seed=5)
    traces = model.fit_batch(list(reads))
    wide = M > 62 or (1 if A <= 2 else 2 if A <= 4 else 3) * M > 64 or K > 8
    assert ("u128" in model.last_sampler) == wide, model.last_sampler
    for u in range(U):
        cfg = orc.make_cfg(K, steps, 2, F, temps, llk_cache_threshold=-1, rng_kind=orc.RNG_PHILOX, seed=5, stream_id=u,
                           break_table=break_table(M, 1.0, 3.0))
        g, l, code = orc.denovo_fit(cfg, reads[u], [A] * M)
        assert code == 0
        assert np.array_equal(traces[u].genotypes, sort_haplotypes(g)), "unit %d" % u
        np.testing.assert_allclose(traces[u].llks, l, rtol=1e-10)


def test_fixed_positions_and_initial_in_a_wide_unit():
    """Homozygous-fixed columns are re-inserted around two-word haplotypes; a caller's `initial` is packed into them."""
    from oracle import binding as orc

    from mchap_amd import DenovoMCMC
    from mchap_amd.assemble import break_table
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    K, M, R = 4, 90, 80
    reads, calls, truth = synth_units(1, ploidy=K, n_pos=M, n_reads=R, first_unit=77, window=(30, 90), qual=(35, 40))
    rd = reads[0].copy()
    for j in (3, 40, 88):  # every read calls allele 0 with high quality: the position is fixed homozygous
        rd[:, j, 0], rd[:, j, 1] = 0.9999, 0.0001 / 3
    model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=20, chains=2, random_seed=9)
    tr = model.fit_batch([rd])[0]
    assert "u128" in model.last_sampler
    cfg = orc.make_cfg(K, 20, 2, None, (1.0,), llk_cache_threshold=-1, rng_kind=orc.RNG_PHILOX, seed=9, stream_id=0,
                       break_table=break_table(M, 1.0, 3.0))
    g, l, code = orc.denovo_fit(cfg, rd, [2] * M)
    assert code == 0 and (g[..., [3, 40, 88]] == 0).all()
    assert np.array_equal(tr.genotypes, sort_haplotypes(g))
    hom = orc.homozygosity_probabilities(rd, [2] * M, K)  # assemble/mcmc.py:168-177: any allele >= fix_homozygous
    n_het = int(M - (np.asarray(hom) >= 0.999).any(axis=-1).sum())
    assert n_het <= M - 3 and n_het > 64
    ini = np.random.default_rng(2).integers(0, 2, size=(2, K, n_het)).astype(np.int8)
    tr2 = model.fit_batch([rd], initial=[ini])[0]
    g2, l2, code = orc.denovo_fit(cfg, rd, [2] * M, initial=ini)
    assert code == 0
    assert np.array_equal(tr2.genotypes, sort_haplotypes(g2))
    np.testing.assert_allclose(tr2.llks, l2, rtol=1e-10)


def test_beyond_the_widest_sampler_is_still_refused_by_name():
    from mchap_amd import DenovoMCMC
    from mchap_amd.synth import synth_units

    reads, _, _ = synth_units(1, ploidy=2, n_pos=127, n_reads=20, window=(30, 127))
    with pytest.raises(NotImplementedError):
        DenovoMCMC(ploidy=2, n_alleles=[2] * 127, steps=5, chains=1, random_seed=1).fit_batch(list(reads))
    reads, _, _ = synth_units(1, ploidy=16, n_pos=4, n_reads=20, window=(2, 4))
    with pytest.raises(NotImplementedError):
        DenovoMCMC(ploidy=16, n_alleles=[2] * 4, steps=5, chains=1, random_seed=1).fit_batch(list(reads))


@pytest.mark.parametrize("case", ["tetraploid-80-snvs", "triploid-50-triallelic", "decaploid-6-snvs", "dodecaploid-wandering", "tetraploid-70-snvs-wandering",
                                  "mixed-ploidies-3-and-11"])
def test_device_summary_of_wide_batches_equals_the_host_classes(case):
    """Round 5: the posterior summary and the replicate incongruence of a batch on the general sampler (two words per haplotype:
    mchap_trace_posterior_*_wph_device, trace_posterior_kernel<32>) against the host classes on the same traces
    (GenotypeMultiTrace.burn().posterior(), mode_genotype_support(), replicate_incongruence(): assemble/classes.py:87-128,
    194-205, 280-376 -- pinned to the reference's vectors in tests/test_classes.py): distinct genotypes in the reference's
    order, probabilities, SPM / GPM, mode genotype, MCI.  Settled units stay with the batch launch; chains of four reads wander
    through more than 512 genotypes and go to the listed launch (ploidy 4: its table holds every state) or beyond it to the
    host classes (ploidy 12: 721 states of 24 words fit the LDS)."""
    from mchap_amd import DenovoMCMC
    from mchap_amd.device import DenovoRaggedBatch
    from mchap_amd.synth import synth_units

    shapes = {"tetraploid-80-snvs": [(4, 80, 60, 2, dict(window=(20, 80)))] * 3, "triploid-50-triallelic": [(3, 50, 40, 3, dict(window=(10, 50)))] * 3,
              "decaploid-6-snvs": [(10, 6, 60, 2, dict(window=(3, 6)))] * 3,
              "dodecaploid-wandering": [(12, 5, 4, 2, dict(window=(2, 5), qual=(2, 8)))] * 2 + [(12, 5, 80, 2, dict(window=(2, 5)))],
              "tetraploid-70-snvs-wandering": [(4, 70, 4, 2, dict(window=(10, 40), qual=(2, 8)))] * 2 + [(4, 70, 60, 2, dict(window=(20, 70)))],
              "mixed-ploidies-3-and-11": [(3, 66, 50, 2, dict(window=(20, 66))), (11, 4, 40, 2, dict(window=(2, 4))), (3, 8, 30, 2, dict(window=(3, 8)))]}[case]
    steps, burn = (600, 100) if "wandering" in case else (120, 40)
    units = []
    for u, (K, M, R, A, skw) in enumerate(shapes):
        reads, _, _ = synth_units(1, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=940 + u, **skw)
        units.append(dict(reads=reads[0], counts=None, n_alleles=[A] * M, ploidy=K, inbreeding=None, stream_id=u))
    model = DenovoMCMC(ploidy=4, n_alleles=[2], steps=steps, chains=2, random_seed=8)
    batch = DenovoRaggedBatch(model, units)
    assert batch.wph == 2
    batch.run(burn)
    got = batch.results()
    n = batch.p_n.cpu().numpy()
    status, fixed = batch.d_status.cpu().numpy(), batch.d_fixed.cpu().numpy()
    for u in range(len(units)):
        want = batch._host_summary(u, int(status[u]), fixed, {"whole": True})
        assert np.array_equal(got[u]["genotypes"], want["genotypes"]), u
        np.testing.assert_array_equal(got[u]["probabilities"], want["probabilities"])
        # (SPM: the device adds a support's probabilities one after the other in ranked order, numpy's sum pairwise: 1e-12 as in
        # tests/test_gpu_posterior.py)
        assert abs(got[u]["spm"] - want["spm"]) < 1e-12 and abs(got[u]["gpm"] - want["gpm"]) < 1e-15 and got[u]["mci"] == want["mci"], (u, got[u]["spm"], want["spm"])
        assert np.array_equal(got[u]["mode_genotype"], want["mode_genotype"])
    if "wandering" in case:
        k = [len(g["probabilities"]) for g in got]
        assert max(k[:2]) > 512 and k[2] <= 512, k          # the listed launch (or the host classes beyond it) did the first two
        assert (n[2] == k[2]) and (case.startswith("dodeca") or (n[:2] == np.array(k[:2])).all()), (n, k)
    else:
        assert (n == [len(g["probabilities"]) for g in got]).all()
    # a part of the units only (the programs' slow records)
    only = batch.results(only=[len(units) - 1])
    assert len(only) == 1 and np.array_equal(only[0]["genotypes"], got[-1]["genotypes"]) and only[0]["mci"] == got[-1]["mci"]
