"""GPU parity of denovo_fill_kernel (the phased sampler's table completion with ONE LANE PER REQUEST: an alternative engine
kept in the parity suite's library, DESIGN.md Appendix A.1, former 4.1d) against the shipped completion inside the exporting launch (the code of a visit: structural.py:433-673 through
denovo_spec_kernel's spec_structural, wave-wide likelihood evaluations).

The tables hold the total move probability of every interval step of a settled chain's genotype; the coasting kernel
compares uniforms with them, so a difference of one unit in the last place would hardly ever show in a trace.  This test
therefore compares the TABLES: every entry of every chain, bit for bit (NaN = not evaluated and -1 = no options included),
on shapes that cover packed and wide request keys, one and several tiles of the read table, partial read chunks, bi- and
tri-allelic positions, inbred priors and shallow units.  Two independently written evaluations of the same likelihoods -- lanes over reads with a wavefront butterfly, and lanes over
requests walking the reads in the butterfly's order -- agreeing to the last bit is also the strongest check there is on
either's summation order."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(reads, flags, **kw):
    """First phase of the phased sampler (steps, table completion, first coasting launch) -> the chains' interval tables."""
    import os

    import torch

    from mchap_amd import DenovoMCMC, _lib
    from mchap_amd.device import DenovoDeviceBatch

    os.environ["MCHAP_HIP_TEST_KERNELS"] = "1"  # mchap_debug_pipe_memo lives in the parity suite's library
    os.environ["MCHAP_HIP_PIPE_STOP"] = "1"
    os.environ["MCHAP_HIP_FLAGS"] = str(flags)
    try:
        model = DenovoMCMC(random_seed=42, kernel=5, **kw)
        b = DenovoDeviceBatch(model, reads)
        assert "phased" in b.sampler_name
        b.run()
        torch.cuda.synchronize()
        U, M = reads.shape[0], reads.shape[2]
        E = M * (M + 1) // 2
        memo = np.zeros((U * model.chains, 2, E), dtype=np.float64)
        f = _lib.lib().mchap_debug_pipe_memo
        f.restype = C.c_int
        rc = f(C.byref(b.cfg), U, _lib.ptr(b.units_host), C.c_void_p(b.d_ws.data_ptr()), _lib.ptr(memo))
        assert rc == 0, _lib.last_error()
        return memo
    finally:
        for k in ("MCHAP_HIP_TEST_KERNELS", "MCHAP_HIP_PIPE_STOP", "MCHAP_HIP_FLAGS"):
            os.environ.pop(k, None)


CASES = {
    # name: (units, ploidy, n_pos, n_reads, n_alleles, inbreeding, synth kwargs)
    "config2": (48, 4, 8, 200, 2, None, {}),
    "config2-inbred": (16, 4, 8, 200, 2, 0.1, {}),
    "diploid": (16, 2, 8, 90, 2, None, {}),
    "triploid-odd-reads": (16, 3, 7, 131, 2, None, dict(window=(3, 7))),
    "hexaploid": (12, 6, 8, 200, 2, None, {}),
    "triallelic-wide-keys": (12, 4, 12, 150, 3, None, dict(window=(4, 12))),
    "shallow": (24, 4, 8, 20, 2, None, dict(qual=(25, 40))),
    "octoploid-deep-tiles": (4, 8, 20, 1000, 2, None, dict(window=(8, 20))),
    "deep-2600-reads": (3, 4, 6, 2600, 2, None, dict(window=(2, 6), qual=(10, 40))),
}


@pytest.mark.parametrize("case", list(CASES))
def test_lane_per_request_tables_equal_the_in_kernel_completion(case):
    from mchap_amd.synth import synth_units

    U, K, M, R, A, F, skw = CASES[case]
    reads, _, _ = synth_units(U, ploidy=K, n_pos=M, n_reads=R, n_alleles=A, first_unit=11, **skw)
    kw = dict(ploidy=K, n_alleles=[A] * M, steps=60, chains=2, inbreeding=F)
    new = _tables(reads, 64, **kw)   # tuning flag 64: denovo_fill_kernel (parity suite's library)
    old = _tables(reads, 0, **kw)    # the shipped path: completion inside the exporting launch
    assert new.shape == old.shape
    done = ~np.isnan(old)
    # (entries beyond the unit's non-fixed positions stay NaN; most chains of these batches settle and get complete tables)
    assert done.any(axis=(1, 2)).mean() > 0.5
    assert np.array_equal(np.isnan(new), np.isnan(old))
    assert np.array_equal(new[done].view(np.uint64), old[done].view(np.uint64)), \
        "max |diff| %.3e" % np.nanmax(np.abs(new - old))
    assert (old[done] >= 0).any() and (old[done] == -1.0).any()  # both kinds of entries occur
