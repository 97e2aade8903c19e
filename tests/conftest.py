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


def free_port() -> int:
    """a rendezvous port the kernel just handed out (the torchrun tests used fixed ports 29517 / 29533, a flake on a
    shared box); bound to 127.0.0.1 and released right before torch.distributed.run takes it"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])
