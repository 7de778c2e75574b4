"""GPU parity, randomised: shapes, priors, tempering ladders and step probabilities drawn at random; every sampler
kernel that supports the shape must reproduce the oracle's trace step for step (tests/fuzz_kernels.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_shapes_match_oracle(seed):
    import fuzz_kernels

    assert fuzz_kernels.run(12, seed) == 0


@pytest.mark.parametrize("seed", [21, 22])
def test_high_ploidy_deep_reads_tempered(seed):
    """The corner the speculative kernel's register allocation is most strained in: K in {6, 8}, more than 256 reads
    (beyond the product-reuse window), always a temperature ladder."""
    import fuzz_kernels

    assert fuzz_kernels.run(6, seed, ploidies=(6, 8), read_depths=(260, 300, 400, 520), tempering=True, max_pos=10) == 0


@pytest.mark.parametrize("seed", [31])
def test_high_ploidy_single_temperature(seed):
    import fuzz_kernels

    assert fuzz_kernels.run(6, seed, ploidies=(6, 8), read_depths=(70, 200, 300), tempering=False, max_pos=12) == 0


@pytest.mark.parametrize("seed", [41, 42, 43])
def test_ragged_launches_match_oracle(seed):
    """Ragged batches as `mchap assemble` builds them (units of different SNV counts, alleles per SNV, depths, qualities,
    priors; 1 to 300 units per launch) through DenovoRaggedBatch's single launch: every unit's trace and device-side
    posterior summary against the oracle (tests/fuzz_ragged.py)."""
    import fuzz_ragged

    assert fuzz_ragged.run(8, seed, verbose=True) == 0
