import os
import sys

import pytest

# PyTorch-ROCm bundles its own HIP/HSA runtime.  Whichever HIP runtime is loaded first serves the whole process, and a
# second one cannot open the GPU any more; tests that use torch on the GPU therefore need torch imported BEFORE
# libfirefly_hip.so pulls in /opt/rocm's runtime (bench.py imports torch first for the same reason).
try:
    import torch  # noqa: F401
except ImportError:  # the product itself does not need torch
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """Build the in-tree artefacts if a fresh checkout has none (the .so files are git-ignored): the HIP library with
    hipcc (cross-compiles without a GPU) and the CPU oracle with gcc.  Same recipe as __graft_entry__.build()."""
    import subprocess
    lib_so = os.path.join(ROOT, "gpupathtracer_amd", "libfirefly_hip.so")
    if not os.path.exists(lib_so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gpupathtracer_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libff_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libff_oracle.so"])


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
def _session_tracer():
    """One Tracer for the whole GPU session (a single process owns the card)."""
    from gpupathtracer_amd import lib
    t = lib.Tracer(0)
    yield t
    t.close()


@pytest.fixture
def tracer(_session_tracer):
    """The session's Tracer.  The library reads its FF_* experiment switches once, at ff_create: a test that flips one calls
    tracer.reload_switches(); whatever it left behind is re-read here from the restored environment (this fixture is set up
    before `monkeypatch` where a test lists it first, so it is torn down after the environment is back)."""
    yield _session_tracer
    _session_tracer.reload_switches()
