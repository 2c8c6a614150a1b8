import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pre3():
    """The product package (directory name starts with a digit -> importlib)."""
    return importlib.import_module("3pre_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def sr4000():
    from util import load_sr4000
    return load_sr4000()
