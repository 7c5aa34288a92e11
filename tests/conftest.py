import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs through the HIP C-ABI)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    from oracle import genjax_oracle as O
    O.build()


@pytest.fixture()
def hostsim():
    """CPU stand-in for libgenmi_hip.so: host-logic tests only (tests/hostsim)."""
    import tests.hostsim as hs
    be = hs.install()
    yield be
    hs.uninstall()


@pytest.fixture()
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from genjax_amd import _lib
    _lib.install(None)
    return _lib.get()
