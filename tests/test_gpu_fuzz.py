"""GPU parity, randomised: shapes, priors, tempering ladders and step probabilities drawn at random; every sampler
kernel that supports the shape must reproduce the oracle's trace step for step (tests/fuzz_kernels.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_match_oracle(seed):
    import fuzz_kernels

    assert fuzz_kernels.run(16, seed) == 0
