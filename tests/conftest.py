import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def fgc_option():
    """fgc_option(name, value): set a process-level library option (fgc_set_option) for the rest of the test; the old value
    comes back at teardown.  (The library reads FGC_* environment variables only once, as initial values.)"""
    from facet_graph_convolution_amd import _lib
    saved = {}

    def setter(name, value):
        if name not in saved:
            saved[name] = _lib.get_option(name)
        _lib.set_option(name, value)

    yield setter
    for name, value in saved.items():
        _lib.set_option(name, value)
