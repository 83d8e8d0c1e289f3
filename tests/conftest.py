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
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def emu_backend():
    """CPU emulation of the HIP kernels (tests/emu): the SAME kernel sources compiled for the host with
    a fiber-based SIMT emulator.  Test-only: points the ctypes binding at it and lets ops accept host
    pointers.  Never used by the product."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("emulation tests are for GPU-less boxes")
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from unipose_amd import _C
    _C.load(build_emu.build())
    _C._ALLOW_HOST_POINTERS = True
    yield torch.device("cpu")
    _C._ALLOW_HOST_POINTERS = False
    _C._lib = None
