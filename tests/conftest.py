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


def pytest_sessionfinish(session, exitstatus):
    """On a GPU box: drain the device BEFORE the interpreter starts tearing modules down.  Everything the tests launched must have finished by
    here -- a kernel that never ends is a test failure, not something to discover as a 'GPU Hang' abort while atexit handlers free memory
    under it (seen once in r04 after a green 56-test run; the suites pass one by one).  The drain is bounded: a watchdog thread turns a device
    that does not come back within two minutes into a loud failure with its own exit status."""
    if not _gpu_available():
        return
    import gc
    import threading
    try:
        import torch
    except Exception:
        return
    if not torch.cuda.is_available():
        return
    done = threading.Event()

    def watchdog():
        if not done.wait(120.0):
            sys.stderr.write("\nFATAL: the GPU did not drain within 120 s after the last test: a kernel launched by the tests never finished\n")
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    gc.collect()                         # banks / pipelines / engines still referenced by collected frames: close them while the runtime is whole
    torch.cuda.synchronize()
    done.set()
