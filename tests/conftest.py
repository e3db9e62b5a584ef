import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/liboracle.so).  Test infrastructure only."""
    from oracle import binding
    return binding.load()


@pytest.fixture(scope="session")
def host_hooks():
    path = os.path.join(ROOT, "tests", "native", "libhost_hooks.so")
    if not os.path.exists(path):
        pytest.fail("tests/native/libhost_hooks.so missing: run ./build.sh")
    return ctypes.CDLL(path)


@pytest.fixture(scope="session")
def gpu_ctx():
    """One bpg context on cuda:0 for the whole GPU session (one process, one device)."""
    import bulletproofs_gadgets_amd as bpg
    ctx = bpg.Context(0)
    yield ctx
    ctx.close()
