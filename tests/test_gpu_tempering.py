"""GPU: temperature ladders with ONE WAVEFRONT PER REPLICA (round 4: denovo_spec_kernel<.., TW>, a workgroup of T wavefronts per
chain; the swap chain of tempering.py:61-151 runs through an LDS exchange area behind workgroup barriers).  Traces against the
oracle step for step, and bit for bit against the one-wavefront form (tuning flag 16384) that walks the replicas one after the
other; ladders longer than eight replicas still take that form."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {
    # name: (units, ploidy, n_pos, n_reads, inbreeding, temperatures, synth kwargs)
    "tetraploid-4-replicas": (6, 4, 8, 120, None, (0.001, 0.01, 0.1, 1.0), {}),
    "diploid-8-replicas": (4, 2, 10, 60, None, tuple(np.linspace(0.125, 1.0, 8)), dict(window=(4, 10))),
    "hexaploid-3-replicas-inbred": (4, 6, 6, 100, 0.2, (0.2, 0.6, 1.0), dict(window=(3, 6))),
    "octoploid-2-replicas-deep": (3, 8, 12, 300, None, (0.5, 1.0), dict(window=(6, 12))),
    "triploid-5-replicas-shallow": (6, 3, 7, 20, 0.0, (0.05, 0.1, 0.3, 0.6, 1.0), dict(window=(3, 7), qual=(10, 30))),
    "tetraploid-9-replicas-one-wavefront": (3, 4, 6, 80, None, tuple(np.linspace(0.2, 1.0, 9)), dict(window=(3, 6))),
}


def _fit(model, reads, flags):
    os.environ["MCHAP_HIP_FLAGS"] = str(flags)
    try:
        return model.fit_batch(list(reads))
    finally:
        os.environ.pop("MCHAP_HIP_FLAGS", None)


@pytest.mark.parametrize("case", list(CASES))
def test_replicas_side_by_side(case):
    from oracle import binding as orc

    from mchap_amd import DenovoMCMC
    from mchap_amd.assemble import break_table
    from mchap_amd.classes import sort_haplotypes
    from mchap_amd.synth import synth_units

    U, K, M, R, F, temps, skw = CASES[case]
    reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, first_unit=300, **skw)
    steps = 80
    model = DenovoMCMC(ploidy=K, n_alleles=[2] * M, steps=steps, chains=2, inbreeding=F, temperatures=temps, random_seed=13)
    side = _fit(model, reads, 0)
    assert "denovo_spec_kernel" in model.last_sampler and "phased" not in model.last_sampler
    one = _fit(model, reads, 16384)  # the replicas one after the other on one wavefront
    for u in range(U):
        assert np.array_equal(side[u].genotypes, one[u].genotypes) and np.array_equal(side[u].llks.view(np.uint64), one[u].llks.view(np.uint64))
        cfg = orc.make_cfg(K, steps, 2, F, temps, llk_cache_threshold=-1, rng_kind=orc.RNG_PHILOX, seed=13, stream_id=u,
                           break_table=break_table(M, 1.0, 3.0))
        g, l, code = orc.denovo_fit(cfg, reads[u], [2] * M)
        assert code == 0
        assert np.array_equal(side[u].genotypes, sort_haplotypes(g)), "unit %d" % u
        np.testing.assert_allclose(side[u].llks, l, rtol=1e-10)
