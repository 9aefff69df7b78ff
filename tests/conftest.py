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
    """The package.  On a fresh checkout the native libraries are compiled first (hipcc cross-compiles without a GPU); the product
    itself never builds or falls back on its own -- a missing libart_hip.so is an error there."""
    import __graft_entry__ as ge
    pkg = ge.load_package()
    if not os.path.exists(pkg.LIB_PATH) or not os.path.exists(os.path.join(pkg.PKG_DIR, "libart_host.so")):
        ge.build()
    return pkg


@pytest.fixture(scope="session")
def backend(art):
    """The GPU backend through the C ABI.  Fails loudly (no fallback) when the library or the GPU is missing."""
    be = art.Backend(0)
    yield be
    be.shutdown()
