import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    return graft.load_oracle()


@pytest.fixture(scope="session")
def aligner(pkg):
    """One engine per session; fails loudly (no CPU fallback) when the GPU or the .so is missing."""
    al = pkg.MI355Aligner(device=0)
    yield al
    al.close()
