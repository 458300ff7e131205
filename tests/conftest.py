import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _library_present():
    """A tree that was never built (fresh clone) gets the product library built once, the
    way __graft_entry__.build() does; an existing build is left alone."""
    from loudgain_amd import build as b
    if not os.path.exists(b.LIB):
        b.build_all()


@pytest.fixture(scope="session")
def oracle():
    from oracle import lgoracle
    lgoracle.lib()
    return lgoracle
