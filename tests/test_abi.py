"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/mchap_hip.h
declares.  No compute calls here."""
import os
import re

import pytest

from tests.conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "mchap_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mchap_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mchap_amd import _lib

    _lib.build()
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.EXPORTS) == names
    assert b"gfx950" in L.mchap_version()


def test_fails_loudly_without_gpu():
    from mchap_amd import DenovoMCMC, _lib
    from mchap_amd.synth import synth_units

    if _lib.lib().mchap_device_count() > 0:
        pytest.skip("a GPU is visible")
    reads, _, _ = synth_units(1, n_reads=10, n_pos=3)
    with pytest.raises(_lib.MchapLibraryError):
        DenovoMCMC(ploidy=4, n_alleles=[2, 2, 2], steps=5, random_seed=1).fit(reads[0])


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/: neither the package nor
    the profiling tools do."""
    bad = []
    for top in ("mchap_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".inc", ".sh")):
                    txt = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "mchap_oracle" in txt or "libmchap_oracle" in txt:
                        bad.append(os.path.join(top, f))
    assert not bad, bad


def test_lds_and_workspace_sizing():
    import ctypes as C
    from mchap_amd import DenovoMCMC, _lib

    L = _lib.lib()
    # config #2: 8 SNVs x 2 alleles x 256 padded reads x 8 B = 32 KiB + per-chain scratch
    n = L.mchap_denovo_lds_bytes(200, 8, 2, 4, 2, 1)
    assert 32768 < n < 40000
    assert L.mchap_denovo_lds_bytes(200, 8, 2, 12, 2, 1) < 0
    cfg = DenovoMCMC(ploidy=4, n_alleles=[2] * 8, random_seed=1, kernel=1)._cfg(8)
    # the call's break table ((8 + 1) x 8 doubles, rounded to 256 B) + 1024 cache entries of 16 B per chain
    assert L.mchap_denovo_workspace_bytes(C.byref(cfg), 10, None) == 768 + 10 * 2 * 1024 * 16
