"""mchap_amd: MI355X-native (gfx950 HIP) implementation of MCHap's per-locus MCMC haplotype
assembler (DenovoMCMC.fit) and exact genotype caller, behind the reference's operator API.

The compute path is the C-ABI shared library built from mchap_amd/csrc (see include/mchap_hip.h);
importing this package does not require a GPU, calling a compute entry point does.
"""
from .assemble import DenovoMCMC  # noqa: F401
from .classes import (  # noqa: F401
    GenotypeMultiTrace,
    GenotypeSupportDistribution,
    PosteriorGenotypeDistribution,
)

__version__ = "0.1.0"
