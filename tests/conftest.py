import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def agx_lib():
    from alphagomoku_amd import build
    build.build(verbose=False)
    from alphagomoku_amd import lib, _lib
    _lib.require_current_build()   # a prebuilt library from other sources must not pass for this tree
    return lib
