import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# PyTorch ships its own HIP runtime with the same soname as the system's.  A process that uses both torch's GPU side and
# librenderer_hip.so must load torch first, so that the renderer binds to the runtime torch already loaded; the other
# order leaves two runtimes in the process and the second one finds no GPU.  Some GPU tests use torch (strip exchange).
try:
    import torch  # noqa: F401,E402
except ImportError:
    pass


# The tests' scenes are small: in the default mode none of their passes would be binned early (that is for passes with
# millions of pairs), so the suite asks for early binning wherever it is valid and keeps covering that path; tests of
# the one-stream path switch it off themselves.
os.environ.setdefault("RH_EARLY_BINNING", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import loader
    loader.build()
    return loader
