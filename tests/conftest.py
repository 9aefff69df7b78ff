import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def art():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def backend(art):
    """The GPU backend through the C ABI.  Fails loudly (no fallback) when the library or the GPU is missing."""
    be = art.Backend(0)
    yield be
    be.shutdown()
