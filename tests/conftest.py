import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a box with a GPU; fails loudly (never skips to a fallback)."""
    from dbot_ros_amd import _capi
    lib = _capi.load()
    assert lib.rbs_device_count() > 0, "no HIP device visible to librbsensor_mi355x.so"
    return lib
