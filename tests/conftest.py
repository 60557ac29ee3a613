import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        from tracklab_amd import _lib
        return os.path.exists(_lib.LIB_PATH) and _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need libtlk.so and an MI355X: on a machine without one they are skipped (a plain `pytest` stays green) instead of
    failing with TlkError 'no HIP device' -- the product itself still fails loudly, it has no CPU fallback."""
    if not any("gpu" in item.keywords for item in items) or _gpu_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP device + tracklab_amd/lib/libtlk.so (run on the GPU box: pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle
