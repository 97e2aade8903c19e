import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))      # tools/lowcx.py: low-complexity / off-model generator of the parity fuzz


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """libccsx.so + the oracle are built once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    return True
