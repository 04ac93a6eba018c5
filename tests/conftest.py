import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(params=["sparse", "dense"])
def vmr_format(request, monkeypatch):
    """Both data layouts of the engine (vmr_data_format): report lists and dense tiles.  Tests that take this
    fixture run once per layout, so the dense tile path is held to the oracle as well as the default one."""
    monkeypatch.setenv("VMR_FORMAT", request.param)
    return request.param
