import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_build():
    """The CPU suite needs the in-tree libraries; build them once if they are missing."""
    import __graft_entry__ as g
    need = [os.path.join(ROOT, "tungsten_amd", "lib", "libtungsten_hip.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        g.build()
