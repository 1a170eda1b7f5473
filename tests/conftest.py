import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import providers
    return providers.oracle()


@pytest.fixture(scope="session")
def ref():
    import providers
    r = providers.ref()
    if r is None:
        pytest.skip("/root/reference not present (GPU box): reference objects cannot be built")
    return r


@pytest.fixture(scope="session")
def mi355():
    import providers
    return providers.mi355()   # raises (fails loudly) when the HIP library or the GPU is missing


@pytest.fixture(scope="session")
def emu():
    import providers
    return providers.emu()
