import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """The C-ABI library is a build product (git-ignored): compile it when a fresh checkout has none, so that the
    suite does not depend on __graft_entry__.build() having run first (hipcc cross-compiles without a GPU)."""
    from mchap_amd import _lib

    if not os.path.exists(_lib.SO) and not os.environ.get("MCHAP_HIP_LIB"):
        _lib.build()
    return _lib.SO
