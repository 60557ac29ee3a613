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


def test_integration_md_indexes_every_entry_point_with_its_header_section():
    """INTEGRATION.md's appendix (tools/gen_abi_index.py) lists every declared entry point next to the header section that cites the reference
    code it replaces, and is regenerated whenever tlk.h changes."""
    import subprocess
    import sys
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    index = text[text.index("abi-index:begin"):text.index("abi-index:end")]
    missing = [s for s in _declared_symbols() if f"`{s}`" not in index]
    assert not missing, f"not in the C ABI index: {missing}"
    assert subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_abi_index.py"), "--check"]).returncode == 0, \
        "INTEGRATION.md's C ABI index is stale: python tools/gen_abi_index.py"


def test_split_fuse_sum_validates_its_arguments_before_touching_a_device():
    """tlk_split_fuse_sum (r06): the shape / term-table checks are host code and come first -- callable without a GPU"""
    from tracklab_amd import _lib
    L = _lib.lib()
    _lib._bind_conv16(L)
    P, I = ctypes.c_void_p * 1, ctypes.c_int * 1
    none, zero = P(None), I(0)
    args = lambda n_terms, c, shift: (n_terms, none, none, none, none, I(shift), zero, 1, 8, 8, c, 0, None, None, 0, None, 0, None)      # noqa: E731
    assert L.tlk_split_fuse_sum(*args(0, 8, 0)) == -1 and b"1..4 terms" in L.tlk_last_error()
    assert L.tlk_split_fuse_sum(*args(5, 8, 0)) == -1
    assert L.tlk_split_fuse_sum(*args(1, 12, 0)) == -1 and b"multiple of 8" in L.tlk_last_error()
    assert L.tlk_split_fuse_sum(*args(1, 8, 0)) == -1 and b"no output" in L.tlk_last_error()
