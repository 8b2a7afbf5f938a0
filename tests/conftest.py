import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Without a GPU every kernel runs on the fiber emulator, where a frame's region growing with the automatic eight
    # wavefronts per frame costs four times the wall clock of two: the CPU suite defaults to two (the tests that are about
    # the multi-wavefront kernel set their own count).  On a GPU box the library's own policy is left alone.
    if "PLH_GROW_MW_WAVES" not in os.environ:
        try:
            import torch
            has_gpu = torch.cuda.is_available()
        except Exception:
            has_gpu = False
        if not has_gpu:
            os.environ["PLH_GROW_MW_WAVES"] = "2"


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
