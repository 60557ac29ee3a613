"""CPU-side checks of the drop-in boundary: libtlk.so loads and exports every symbol declared in
include/tlk.h (no compute calls here: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "tlk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tlk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tracklab_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, f"declared in tlk.h but not exported: {missing}"


def test_version_and_error_string():
    from tracklab_amd import _lib
    L = _lib.lib()
    assert L.tlk_version() >= 100
    assert isinstance(L.tlk_last_error(), bytes)


def test_no_cpu_fallback_without_device():
    """On a box without a GPU, creating a tracker must fail loudly (never fall back to CPU)."""
    from tracklab_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.TlkError):
        _lib.OCSortBank(0.0)
