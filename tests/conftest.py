import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import _util
    O = _util.oracle()
    O.build()
    return O


@pytest.fixture(scope="session")
def synth():
    import _util
    return _util.synth()


@pytest.fixture(scope="session")
def plslam():
    import _util
    return _util.plslam()


@pytest.fixture(scope="session")
def emu_lib():
    """CPU emulation build of the HIP sources (debug aid, not parity evidence)."""
    import _util
    sys.path.insert(0, _util.ROOT)
    import __graft_entry__ as g
    return g.build_emu()
