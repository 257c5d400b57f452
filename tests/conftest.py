import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    from asr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.load()


@pytest.fixture(scope="session")
def dev(lib):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
