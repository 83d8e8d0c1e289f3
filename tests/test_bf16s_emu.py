"""bf16 STORAGE (BASELINE configs[4]) on the CPU emulator: the same kernel sources, bf16 tensors in "HBM"."""
import pytest

import bf16s_cases as bc

CONVS = [
    # n, c,  h,  w,  k, r, stride, pad, dil, bias
    (2, 64, 6, 5, 32, 1, 1, 0, 1, False),        # 1x1
    (1, 64, 7, 7, 72, 3, 1, 1, 1, False),        # 3x3, K = 72 -> 96 physical channels, ragged N tile
    (1, 64, 7, 7, 64, 3, 1, 3, 3, False),        # dilated, taps in the padding
    (2, 64, 9, 9, 64, 3, 2, 1, 1, False),        # stride 2 (data gradient through the parity gather)
    (1, 128, 5, 5, 64, 1, 1, 0, 1, False),       # several K slices per tap
    (2, 256, 6, 6, 17, 1, 1, 0, 1, True),        # the network's last layer: K = 17 -> 32, bias
    (3, 32, 23, 23, 48, 3, 1, 18, 18, False),    # WASP geometry; K = 48 -> 64 (decoder.conv1's width)
    (2, 160, 7, 7, 136, 3, 1, 1, 1, False),      # two row tiles x several column tiles in the weight gradient
]


@pytest.mark.parametrize("cfg", CONVS)
def test_conv_bf16_storage(emu_backend, cfg):
    bc.conv_case(emu_backend, *cfg)


@pytest.mark.parametrize("cfg", [
    (3, 32, 9, 9, 48, 3, 1, 1, 1, True, False, True),       # pad channels of y / z / dy stay zero (K = 48 -> 64)
    (3, 64, 9, 9, 64, 3, 2, 1, 1, True, True, True),        # residual + ReLU
    (4, 64, 5, 5, 96, 1, 1, 0, 1, False, True, True),       # downsample-like: no ReLU, residual
    (2, 32, 7, 7, 32, 3, 1, 2, 2, True, False, False),      # eval: BatchNorm folded into the convolution epilogue
])
def test_conv_bn_bf16_storage(emu_backend, cfg):
    n, c, h, w, k, r, s, p, d, relu, residual, train = cfg
    bc.conv_bn_case(emu_backend, n, c, h, w, k, r, s, p, d, relu=relu, residual=residual, train=train)


def test_small_ops_bf16_storage(emu_backend):
    bc.small_ops_case(emu_backend)


def test_last_convolution_writes_fp32(emu_backend):
    bc.f32_out_case(emu_backend)
    bc.f32_out_case(emu_backend, n=1, c=96, h=7, w=6, k=22, r=3, pad=1, seed=9)


def test_model_eval_bf16_storage(emu_backend):
    bc.model_eval_case(emu_backend)


def test_model_train_bf16_storage(emu_backend):
    bc.model_train_case(emu_backend)


def test_tile_rule_of_the_bf16_kernels_and_statistics_rows(emu_backend):
    """The plain-bf16 kernels choose their tile with their own threshold ("tile_want_bf16"), so the BatchNorm partial rows
    the convolution epilogue writes follow up_conv_stats_tiles_math, not the fp32 rule: force the two rules apart (fp32:
    the largest tile, bf16: the smallest) and run a training conv+BN in bf16 storage over several row tiles."""
    from unipose_amd import _C
    lib = _C.lib()
    try:
        _C.check(lib.up_conv_tune(b"tile_want", 1), "tile_want")
        _C.check(lib.up_conv_tune(b"tile_want_bf16", 100000), "tile_want_bf16")
        bc.conv_bn_case(emu_backend, 3, 32, 9, 9, 48, 3, 1, 1, 1, relu=True, residual=False, train=True)   # 243 rows: 2 vs 4 tiles
        _C.check(lib.up_conv_tune(b"tile_want", 100000), "tile_want")
        _C.check(lib.up_conv_tune(b"tile_want_bf16", 1), "tile_want_bf16")
        bc.conv_bn_case(emu_backend, 3, 32, 9, 9, 48, 3, 1, 1, 1, relu=True, residual=False, train=True)
    finally:
        lib.up_conv_tune(b"tile_want", 1500)
        lib.up_conv_tune(b"tile_want_bf16", 500)


import glds_cases as gc


@pytest.mark.parametrize("case", gc.SMALL, ids=lambda c: "c%d_%dx%d_k%d_r%d_d%d_t%d" % (c["c"], c["h"], c["w"], c["k"], c["r"], c["dil"], c["tile_want"]))
def test_glds_kernel_matches_register_staged_kernel(emu_backend, case):
    """second-generation bf16-storage kernels (direct-to-LDS loads, tap skipping, tap-sorted rows, 16-byte stores) == the
    register-staged kernels, element for element: outputs, BatchNorm partials, data gradients, weight gradients"""
    gc.conv_ab(emu_backend, **case)


@pytest.mark.parametrize("case", gc.BNRED, ids=lambda c: "c%d_%dx%d_k%d_r%d_t%d_%s" % (c["c"], c["h"], c["w"], c["k"], c["r"], c["tile_want"], "m" if c.get("mask_add") else ("a" if c.get("add") else "n")))
def test_bn_backward_reduction_fused_into_data_gradient_bf16(emu_backend, case):
    gc.bnred_case(emu_backend, **case)


@pytest.mark.parametrize("case", gc.BIG_SMALL, ids=lambda c: "c%d_%dx%d_k%d_r%d_d%d_s%d" % (c["c"], c["h"], c["w"], c["k"], c["r"], c["dil"], c["stride"]))
def test_big_tile_kernel_matches_glds_kernel(emu_backend, case):
    """third-generation bf16-storage kernel (8 waves on (32 TM) x 256 tiles, bf16s_big.h) == igemm_glds_kernel, element for
    element (outputs, data gradients); BatchNorm partials merged to fp32 round-off"""
    gc.conv_ab(emu_backend, **case)


@pytest.mark.parametrize("case", gc.BIG_BNRED, ids=lambda c: "c%d_%dx%d_k%d_r%d_%s" % (c["c"], c["h"], c["w"], c["k"], c["r"], "m" if c.get("mask_add") else ("a" if c.get("add") else "n")))
def test_bn_backward_reduction_fused_into_big_tile_data_gradient(emu_backend, case):
    gc.bnred_case(emu_backend, **case)
