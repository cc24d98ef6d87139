import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine_cls():
    from multifidelity_datafusion_gps_amd._lib import Engine
    return Engine


@pytest.fixture()
def engine(engine_cls):
    """A fresh HIP engine handle; raises (test ERROR) when libmfgp_hip.so or the GPU is missing."""
    e = engine_cls(0)
    yield e
    e.close()
