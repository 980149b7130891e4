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


@pytest.fixture(scope="session")
def gpu():
    """Initialise libmsmhip on device 0; a missing library or GPU is a FAILURE here, not a skip:
    the -m gpu tier must never pass on a fallback."""
    from msmbuilder_amd import _lib
    assert _lib.device_count() >= 1, "no HIP device visible to libmsmhip"
    _lib.ensure_device(0)
    return _lib
