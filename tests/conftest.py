import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def ff():
    """The product library's ctypes binding (host-side functions work without a GPU)."""
    from gpupathtracer_amd import lib
    lib.load()
    return lib


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def tracer():
    """One Tracer for the whole GPU session (a single process owns the card)."""
    from gpupathtracer_amd import lib
    t = lib.Tracer(0)
    yield t
    t.close()
