import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _serialise_make()


def _serialise_make():
    """Tests (re)build their checkers with `make` on demand.  Under pytest-xdist several workers would relink the same binary while another
    one runs it ("Text file busy" / a half-written library): every `make` a test starts takes one lock file first."""
    import fcntl
    import subprocess
    real_run = subprocess.run
    lock_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_emu", ".make.lock")

    def run(args, *a, **kw):
        if isinstance(args, (list, tuple)) and args and args[0] == "make":
            os.makedirs(os.path.dirname(lock_path), exist_ok=True)
            with open(lock_path, "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                try:
                    return real_run(args, *a, **kw)
                finally:
                    fcntl.flock(lk, fcntl.LOCK_UN)
        return real_run(args, *a, **kw)
    if getattr(subprocess.run, "__name__", "") != "run" or subprocess.run is real_run:
        subprocess.run = run


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
