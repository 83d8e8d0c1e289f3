"""The persistent stream-K form of the forward / data-gradient kernel (`up_conv_set_persistent`, default off): same
parity cases as the default form, with grids that cut tiles in many different places.  Last file of the suite on
purpose: it toggles a library-wide switch."""
import pytest
import torch

import model_cases as mc
import op_cases as oc

CONV_CASES = [
    # n, c,  h,  w,  k, r, stride, pad, dil, bias, relu
    (2, 16, 6, 5, 32, 1, 1, 0, 1, False, False),      # 1x1, one K slice per tile: boundaries snap to tiles
    (1, 32, 9, 8, 16, 3, 1, 1, 1, False, False),
    (2, 16, 9, 9, 24, 3, 2, 1, 1, False, False),      # stride 2: four parity-class launches in the data gradient
    (1, 3, 20, 18, 8, 7, 2, 3, 1, False, False),      # generic (unaligned) K slices
    (1, 15, 6, 6, 14, 3, 1, 1, 1, True, True),
    (1, 128, 5, 5, 32, 3, 1, 1, 1, False, False),     # double-buffered loop
    (2, 96, 6, 6, 72, 3, 1, 1, 1, False, False),      # 27 slices, two n-tiles, ragged rows
    (1, 2048, 3, 3, 16, 1, 1, 0, 1, True, True),      # 64 slices in ONE tile: pure split-K, bias + ReLU after the merge
    (1, 32, 5, 5, 16, 3, 1, 6, 6, False, False),      # tap skipping leaves one slice: empty shares
    (2, 64, 9, 9, 32, 3, 1, 6, 6, True, True),        # tiles keep different tap subsets (shares scale per tile)
    (2, 16, 13, 11, 80, 3, 1, 1, 1, False, False),    # 10 tiles: workgroups walk several tiles
]


@pytest.fixture
def persistent():
    from unipose_amd import _C

    def set_(grid):
        _C.check(_C.lib().up_conv_set_persistent(1, grid), "conv_set_persistent")
    yield set_
    _C.check(_C.lib().up_conv_set_persistent(0, 0), "conv_set_persistent")


def _ops_suite(dev, cases):
    for n, c, h, w, k, r, s, p, d, bias, relu in cases:
        oc.conv_case(dev, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)
    oc.conv_bn_case(dev, 2, 128, 5, 5, 48, 3, 1, 1, 1, relu=True, residual=True, train=True)   # statistics of merged tiles
    oc.dgrad_add_case(dev, 1, 128, 6, 6, 64, 3, 1, 1, 1)                                       # addend after the merge


@pytest.mark.parametrize("grid", [0, 1, 3, 7, 16])
def test_persistent_ops_emu(emu_backend, persistent, grid):
    persistent(grid)
    _ops_suite(emu_backend, CONV_CASES)


def test_persistent_setter(emu_backend):
    from unipose_amd import _C
    lib = _C.lib()
    assert lib.up_conv_get_persistent() == 0          # default: off
    assert lib.up_conv_set_persistent(1, -1) != 0
    assert lib.up_conv_set_persistent(1, 5) == 0 and lib.up_conv_get_persistent() == 1
    assert lib.up_conv_set_persistent(0, 0) == 0 and lib.up_conv_get_persistent() == 0


def test_persistent_model_emu(emu_backend, persistent):
    persistent(5)
    mc.train_case(emu_backend, size=32)      # forward output, loss, every parameter gradient, running statistics


# ---- on the MI355X: real layer shapes, the library's own grid and two overrides ------------------------------
GPU_CASES = [
    (4, 256, 23, 23, 256, 3, 1, 1, 1, False, False),     # layer3 3x3
    (4, 1024, 23, 23, 256, 1, 1, 0, 1, False, False),    # layer3 1x1 reduce
    (2, 256, 46, 46, 256, 3, 2, 1, 1, False, False),     # layer3.0 stride 2
    (4, 256, 23, 23, 256, 3, 1, 18, 18, False, False),   # WASP d = 18: heavy tap skipping
    (2, 512, 23, 23, 512, 3, 1, 4, 4, False, False),     # layer4 dilated
    (2, 3, 96, 96, 64, 7, 2, 3, 1, False, False),        # stem (generic path)
    (2, 256, 46, 46, 17, 1, 1, 0, 1, True, False),       # output layer with bias
]


@pytest.mark.gpu
@pytest.mark.parametrize("grid", [0, 97, 1500])
def test_persistent_ops_gpu(persistent, grid):
    persistent(grid)
    dev = torch.device("cuda:0")
    _ops_suite(dev, GPU_CASES)


@pytest.mark.gpu
def test_persistent_model_gpu(persistent):
    persistent(0)
    dev = torch.device("cuda:0")
    assert mc.eval_case(dev, size=160) < 1e-3
    mc.train_case(dev, size=96, B=4)
