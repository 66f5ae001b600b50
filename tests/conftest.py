import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_DEVICES = None


def _device_count():
    """HIP devices libasx.so can see; a missing / unloadable library is NOT a reason to skip (the GPU tests must fail
    loudly then), so that case reports one device and lets the tests raise."""
    global _DEVICES
    if _DEVICES is None:
        try:
            from audio_separator_amd.engine import load_library
            _DEVICES = int(load_library().asx_device_count())
        except Exception:
            _DEVICES = 1
    return _DEVICES


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items or _device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible (libasx.so loaded, asx_device_count() == 0)")
    for it in gpu_items:
        it.add_marker(skip)
