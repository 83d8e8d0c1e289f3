"""Two forms that stop multiplying structural zeros of dilated / padded convolutions (both ON by default since the round-2 A/B,
profiles/r02_a_knob_ab.txt; the plain forms stay reachable through up_conv_tune and are covered here too):
* "tap_sort": GEMM rows of the forward / data-gradient kernel ordered by their set of live filter taps, so that the
  tile-level tap skipping becomes near exact;
* "wgrad_rect": the weight-gradient reduction of a column tile runs over the live rectangle of its filter taps only.
Parity with the knobs on, and the host-side predictions (`up_conv_tap_visits`, `up_conv_wgrad_visits`)."""
import ctypes

import pytest
import torch

import model_cases as mc
import op_cases as oc

CASES = [
    # n, c,  h,  w,  k, r, stride, pad, dil, bias, relu
    (1, 32, 9, 8, 16, 3, 1, 1, 1, False, False),      # plain padded 3x3: border classes
    (1, 128, 5, 5, 32, 3, 1, 1, 1, False, False),     # double-buffered loop
    (2, 96, 6, 6, 72, 3, 1, 1, 1, False, False),      # K-split tail tiles on top of the permutation
    (1, 32, 5, 5, 16, 3, 1, 6, 6, False, False),      # dilation > H: a single class (no permutation is built)
    (2, 64, 9, 9, 32, 3, 1, 6, 6, True, True),        # dilation 6 on 9x9: nine classes, bias + ReLU
    (3, 32, 23, 23, 32, 3, 1, 18, 18, False, True),   # the WASP geometry: several tiles per class, three images
    (2, 32, 12, 10, 32, 3, 1, 2, 2, True, False),     # rectangular map
    (2, 32, 11, 11, 32, 5, 1, 2, 1, False, False),    # 5x5 (25 taps > 16: no tap map, rows still permuted? no: image order)
    (2, 16, 9, 9, 24, 3, 1, 1, 1, False, False),      # 16 channels: generic path, knob has no effect
]


@pytest.fixture
def tap_sort():
    from unipose_amd import _C
    _C.check(_C.lib().up_conv_tune(b"tap_sort", 1), "conv_tune")
    yield
    _C.check(_C.lib().up_conv_tune(b"tap_sort", 1), "conv_tune")


@pytest.fixture
def wgrad_rect():
    from unipose_amd import _C
    _C.check(_C.lib().up_conv_tune(b"wgrad_rect", 1), "conv_tune")
    yield
    _C.check(_C.lib().up_conv_tune(b"wgrad_rect", 1), "conv_tune")


@pytest.fixture
def plain_forms():
    """Both switches off: rows in image order, weight-gradient reduction over every pixel."""
    from unipose_amd import _C
    for key in (b"tap_sort", b"wgrad_rect"):
        _C.check(_C.lib().up_conv_tune(key, 0), "conv_tune")
    yield
    for key in (b"tap_sort", b"wgrad_rect"):
        _C.check(_C.lib().up_conv_tune(key, 1), "conv_tune")


WGRAD_CASES = CASES + [
    (2, 16, 9, 9, 24, 3, 2, 1, 1, False, False),      # stride 2: rectangles in output coordinates
    (1, 3, 20, 18, 8, 7, 2, 3, 1, False, False),      # stem 7x7 stride 2: many taps per column tile (bounding rectangle)
    (1, 15, 12, 12, 8, 11, 1, 5, 1, True, True),      # 11x11
    (1, 16, 7, 7, 16, 3, 1, 4, 4, False, False),      # dilation > H/2: corner taps see a 3x3 patch
    (2, 32, 8, 8, 20, 1, 2, 0, 1, False, False),      # 1x1 stride 2: one full rectangle (plain form is kept)
    (1, 32, 9, 9, 16, 3, 2, 0, 1, False, False),      # no padding: every tap sees everything
    (1, 32, 4, 4, 16, 3, 1, 5, 5, False, False),      # dilation > H: eight taps never live (empty rectangles)
]


def _wgrad_visits(n, h, w, c, k, r, dil, stride=1):
    from unipose_amd import _C
    pad = (r // 2) * dil
    po = (h + 2 * pad - dil * (r - 1) - 1) // stride + 1
    qo = (w + 2 * pad - dil * (r - 1) - 1) // stride + 1
    d = _C.ConvDesc(n, h, w, c, c, c, k, r, r, stride, pad, dil, po, qo, k, k)
    f = ctypes.c_double()
    _C.check(_C.lib().up_conv_wgrad_visits(ctypes.byref(d), ctypes.byref(f)), "conv_wgrad_visits")
    return f.value


def test_plain_forms_ops_emu(emu_backend, plain_forms):
    for n, c, h, w, k, r, s, p, d, bias, relu in CASES[:6] + WGRAD_CASES[-7:-4]:
        oc.conv_case(emu_backend, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)


@pytest.mark.gpu
def test_plain_forms_ops_gpu(plain_forms):
    dev = torch.device("cuda:0")
    for n, c, h, w, k, r, s, p, d, bias, relu in CASES + WGRAD_CASES[-7:]:
        oc.conv_case(dev, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)
    oc.conv_case(dev, 4, 256, 23, 23, 256, 3, 1, 18, 18)
    assert mc.eval_case(dev, size=368, B=1, K=16, tol=1e-4) < 1e-4


def test_wgrad_rect_ops_emu(emu_backend, wgrad_rect):
    for n, c, h, w, k, r, s, p, d, bias, relu in WGRAD_CASES:
        oc.conv_case(emu_backend, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)


def test_wgrad_rect_model_emu(emu_backend, wgrad_rect, tap_sort):
    mc.train_case(emu_backend, size=32)                  # both knobs together, every parameter gradient


def test_wgrad_visit_prediction(emu_backend):
    """One tap per column tile (Cp >= 128): the rectangle is exact, i.e. the live share of SURVEY 8d."""
    for dil, live in ((18, 0.229), (12, 0.425), (6, 0.682)):
        assert abs(_wgrad_visits(32, 23, 23, 256, 256, 3, dil) - live) < 2e-3
    assert abs(_wgrad_visits(32, 23, 23, 256, 256, 3, 1) - (21 * 21 * 9 + 84 * 6 + 4 * 4) / (529 * 9.0)) < 1e-9
    assert _wgrad_visits(32, 23, 23, 256, 256, 1, 1) == 1.0
    assert 0.98 < _wgrad_visits(32, 92, 92, 64, 64, 3, 1) < 1.0      # two taps per 128-column tile: bounding rectangle


def _suite(dev, cases):
    for n, c, h, w, k, r, s, p, d, bias, relu in cases:
        oc.conv_case(dev, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)
    oc.conv_bn_case(dev, 2, 32, 9, 9, 48, 3, 1, 2, 2, relu=True, residual=False, train=True)   # statistics of permuted tiles
    oc.conv_bn_case(dev, 2, 128, 5, 5, 48, 3, 1, 1, 1, relu=True, residual=True, train=True)   # residual: image order
    oc.dgrad_add_case(dev, 1, 128, 6, 6, 64, 3, 1, 1, 1)                                       # addend: image order


def test_tap_sort_ops_emu(emu_backend, tap_sort):
    _suite(emu_backend, CASES)


def test_tap_sort_model_emu(emu_backend, tap_sort):
    assert mc.eval_case(emu_backend, size=64, B=1) < 1e-4       # (the train step runs with both knobs below)


def _visits(n, h, w, c, k, r, dil, dgrad=0):
    from unipose_amd import _C
    pad = (r // 2) * dil
    d = _C.ConvDesc(n, h, w, c, c, c, k, r, r, 1, pad, dil, h, w, k, k)
    a, b, live = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _C.check(_C.lib().up_conv_tap_visits(ctypes.byref(d), dgrad, ctypes.byref(a), ctypes.byref(b), ctypes.byref(live)),
             "conv_tap_visits")
    return a.value, b.value, live.value


def test_tap_visit_prediction(emu_backend):
    """BASELINE configs[1] shapes: sorted rows visit (nearly) only live taps; image order keeps most dead ones."""
    for dil, live_want in ((18, 0.229), (12, 0.425), (6, 0.682)):            # 23 %, 43 %, 68 %
        for dgrad in (0, 1):
            img, srt, live = _visits(32, 23, 23, 256, 256, 3, dil, dgrad)
            assert abs(live - live_want) < 2e-3
            assert live <= srt <= live + 0.01 and srt < img
    img, srt, live = _visits(32, 23, 23, 256, 256, 3, 18)
    assert img > 2.2 * srt                                                   # 0.54 -> 0.23 of the K loop
    img, srt, live = _visits(32, 23, 23, 256, 256, 3, 1)                      # layer3 3x3: the border taps, 5.7 % of the MACs
    assert img == 1.0 and abs(live - (21 * 21 * 9 + 84 * 6 + 4 * 4) / (529 * 9.0)) < 1e-9 and srt < 0.95
    img, srt, live = _visits(32, 23, 23, 256, 256, 1, 1)                      # 1x1: one tap, nothing to skip
    assert img == srt == live == 1.0


@pytest.mark.gpu
def test_tap_sort_ops_gpu(tap_sort):
    dev = torch.device("cuda:0")
    _suite(dev, [(4, 256, 23, 23, 256, 3, 1, 18, 18, False, False), (4, 256, 23, 23, 256, 3, 1, 6, 6, False, True),
                 (4, 256, 23, 23, 256, 3, 1, 1, 1, False, False), (2, 512, 23, 23, 512, 3, 1, 8, 8, False, False),
                 (2, 128, 46, 46, 128, 3, 1, 1, 1, True, True)])


@pytest.mark.gpu
def test_wgrad_rect_gpu(wgrad_rect, tap_sort):
    dev = torch.device("cuda:0")
    for cfg in [(4, 256, 23, 23, 256, 3, 1, 18, 18, False, False), (4, 256, 23, 23, 256, 3, 1, 1, 1, False, False),
                (2, 512, 23, 23, 512, 3, 1, 8, 8, False, False), (2, 64, 92, 92, 64, 3, 1, 1, 1, False, False),
                (2, 128, 92, 92, 128, 3, 2, 1, 1, False, False), (2, 3, 96, 96, 64, 7, 2, 3, 1, False, False)]:
        n, c, h, w, k, r, s, p, d, bias, relu = cfg
        oc.conv_case(dev, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)
    mc.train_case(dev, size=96, B=2)


@pytest.mark.gpu
def test_tap_sort_model_gpu(tap_sort):
    dev = torch.device("cuda:0")
    assert mc.eval_case(dev, size=368, B=1, K=16, tol=1e-4) < 1e-4           # 23x23 top maps: the real WASP geometry
    mc.train_case(dev, size=96, B=2)


def test_random_knob_combinations_emu(emu_backend):
    """Random geometries x random combinations of every run-time kernel switch (tap_sort, wgrad_rect, tail_split, and the
    three direct-to-LDS fp32 switches): forward, data gradient, weight gradient and BatchNorm statistics stay within the
    operator tolerances.  (The same generator ran over several hundred cases when the switches were written.)"""
    import random
    from unipose_amd import _C
    lib = _C.lib()
    rnd = random.Random(20260927)
    try:
        done = 0
        while done < 24:
            c, k = rnd.choice([32, 64, 96, 16]), rnd.choice([16, 24, 32, 72])
            r = rnd.choice([1, 3, 3, 3, 5])
            stride = rnd.choice([1, 1, 1, 2])
            dil = rnd.choice([1, 1, 2, 3, 6]) if r > 1 else 1
            pad = rnd.choice([0, (r // 2) * dil, (r // 2) * dil]) if r > 1 else 0
            h, w, n = rnd.randint(4, 16), rnd.randint(4, 16), rnd.randint(1, 3)
            if h + 2 * pad < dil * (r - 1) + 1 or w + 2 * pad < dil * (r - 1) + 1:
                continue
            for key, val in (("tap_sort", rnd.randint(0, 1)), ("wgrad_rect", rnd.randint(0, 1)),
                             ("tail_split", rnd.randint(0, 1)), ("glds32", rnd.randint(0, 1)), ("glds32_epi", rnd.randint(0, 1)),
                             ("glds32_wgrad", rnd.randint(0, 1))):
                _C.check(lib.up_conv_tune(key.encode(), val), key)
            if done % 3 == 2 and r == 3 and stride == 1:
                oc.conv_bn_case(emu_backend, n, c, h, w, k, r, stride, dil, dil, relu=rnd.random() < 0.5,
                                residual=rnd.random() < 0.4, train=True, seed=done)
            else:
                oc.conv_case(emu_backend, n, c, h, w, k, r, stride, pad, dil, bias=rnd.random() < 0.3,
                             relu=rnd.random() < 0.3, seed=done)
            done += 1
    finally:
        for key, val in (("tap_sort", 1), ("wgrad_rect", 1), ("tail_split", 1), ("glds32", 1), ("glds32_epi", 1), ("glds32_wgrad", 1)):
            lib.up_conv_tune(key.encode(), val)
