"""Kernel index/tiling/fragment logic on a GPU-less box: the HIP sources compiled against the fiber
emulator (tests/emu) and driven through the real host code (unipose_amd.ops) with tiny shapes."""
import pytest
import torch

import op_cases as oc


@pytest.mark.parametrize("cfg", [
    # n, c,  h,  w,  k, r, stride, pad, dil, bias, relu
    (2, 16, 6, 5, 32, 1, 1, 0, 1, False, False),      # 1x1 (K1)
    (1, 32, 9, 8, 16, 3, 1, 1, 1, False, False),      # 3x3 (K3)
    (2, 16, 9, 9, 24, 3, 2, 1, 1, False, False),      # 3x3 stride 2 (K4), K=24 -> n-tile tail
    (1, 16, 7, 7, 16, 3, 1, 4, 4, False, False),      # dilated, dil > H/2 (K5: taps wholly in padding)
    (2, 32, 8, 8, 20, 1, 2, 0, 1, False, False),      # 1x1 stride 2 downsample (K2)
    (1, 3, 20, 18, 8, 7, 2, 3, 1, False, False),      # stem 7x7 s2, Cin=3 (K6, unaligned K slices)
    (1, 15, 6, 6, 14, 3, 1, 1, 1, True, True),        # LSTM-like: odd channels, bias, relu
    (1, 15, 12, 12, 8, 11, 1, 5, 1, True, True),      # 11x11 (K17)
    (2, 32, 5, 5, 14, 1, 1, 0, 1, True, False),       # K=14: data gradient with a ragged (16 < 32) K slice
    (1, 128, 5, 5, 32, 3, 1, 1, 1, False, False),     # reduction 1152 >= 1024: double-buffered LDS loop
    (1, 128, 6, 6, 128, 3, 2, 1, 1, False, False),    # same for the strided data gradient
    (2, 32, 7, 7, 16, 1, 2, 0, 1, False, False),      # 1x1 stride 2 on an odd size: three of the four parity classes empty
    (1, 15, 8, 7, 14, 3, 2, 1, 1, True, False),       # stride-2 parity classes on the generic (unaligned) path
    (1, 4, 10, 9, 8, 7, 2, 3, 1, False, False),       # 7x7 stride 2 pad 3: classes of 4x4, 4x3, 3x4, 3x3 taps
    (1, 32, 9, 9, 16, 3, 2, 0, 1, False, False),      # stride 2 without padding
    (1, 32, 5, 5, 16, 3, 1, 6, 6, False, False),      # dilation > H: only the centre tap survives the tap skipping; the
                                                      # K split (4 parts of a 1-slice loop) leaves three parts empty
    (2, 64, 9, 9, 32, 3, 1, 6, 6, True, True),        # dilation 6 on 9x9: tiles keep different tap subsets
])
def test_conv_fwd_bwd(emu_backend, cfg):
    n, c, h, w, k, r, s, p, d, bias, relu = cfg
    oc.conv_case(emu_backend, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)


@pytest.mark.parametrize("cfg,parts", [
    # short tile tail (tiles % CUs small) -> the tail tiles are split along K into `parts` (include/unipose_hip.h):
    # one part per CU, at least 2 K slices each
    ((1, 64, 5, 5, 32, 3, 1, 1, 1, False, False), 9),     # K = 576: 18 slices -> 9 parts of 2
    ((2, 96, 6, 6, 72, 3, 1, 1, 1, False, False), 13),    # 27 slices -> 13 uneven parts; 2 n-tiles, ragged rows
    ((1, 2048, 3, 3, 16, 1, 1, 0, 1, True, True), 32),    # 64 slices -> 32 parts, bias + ReLU epilogue after the merge
    ((1, 256, 6, 6, 64, 3, 2, 1, 1, False, False), 36),   # strided: MODE 1 data gradient is split as well
])
def test_conv_tail_split(emu_backend, cfg, parts):
    import ctypes
    from unipose_amd import _C, ops
    n, c, h, w, k, r, s, p, d, bias, relu = cfg
    x = torch.zeros(n, h, w, c)
    desc = ops.make_desc(x, torch.zeros(k, c, r, r), ops.ConvCfg(s, p, d))
    assert _C.lib().up_conv_split_parts(ctypes.byref(desc)) == parts
    oc.conv_case(emu_backend, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)


@pytest.mark.parametrize("cfg", [
    (2, 32, 6, 5, 16, 1, 1, 0, 1),       # 1x1: the residual-block case
    (1, 128, 6, 6, 64, 3, 1, 1, 1),      # K-split tail tile: the addend joins after the merge
    (1, 14, 6, 6, 15, 3, 1, 1, 1),       # odd channels: pad channel of dx stays 0
])
def test_dgrad_with_addend(emu_backend, cfg):
    oc.dgrad_add_case(emu_backend, *cfg)


def test_conv_bn_tail_split(emu_backend):
    # batch statistics are reduced from the MERGED accumulators of a split tile
    oc.conv_bn_case(emu_backend, 2, 128, 5, 5, 48, 3, 1, 1, 1, relu=True, residual=True, train=True)


def test_conv_multi_tile(emu_backend):
    # M = 2*13*11 = 286 rows -> several 64-row tiles with a ragged tail; K = 80 -> 2 n-tiles of 64
    oc.conv_case(emu_backend, 2, 16, 13, 11, 80, 3, 1, 1, 1)


@pytest.mark.parametrize("cfg", [
    # n, c, h, w, k, r, stride, pad, dil, relu, residual, train
    (2, 16, 6, 6, 32, 1, 1, 0, 1, True, False, True),
    (2, 16, 7, 7, 16, 3, 1, 2, 2, True, True, True),
    (3, 8, 9, 9, 72, 3, 2, 1, 1, False, False, True),
    (2, 16, 5, 5, 16, 1, 1, 0, 1, True, True, False),     # eval statistics, with grad (freeze_bn path)
])
def test_conv_bn_act(emu_backend, cfg):
    n, c, h, w, k, r, s, p, d, relu, res, train = cfg
    oc.conv_bn_case(emu_backend, n, c, h, w, k, r, s, p, d, relu=relu, residual=res, train=train)


def test_layout(emu_backend):
    oc.layout_case(emu_backend)


def test_maxpool(emu_backend):
    oc.maxpool_case(emu_backend)
    oc.maxpool_case(emu_backend, 1, 4, 8, 8)


@pytest.mark.parametrize("shape", [(2, 8, 5, 6, 10, 12), (1, 4, 1, 1, 7, 7), (1, 4, 4, 4, 32, 32), (1, 4, 6, 6, 6, 6)])
def test_bilinear(emu_backend, shape):
    oc.bilinear_case(emu_backend, *shape)


def test_gap(emu_backend):
    oc.gap_case(emu_backend)


def test_concat(emu_backend):
    oc.concat_case(emu_backend)


def test_dropout(emu_backend):
    oc.dropout_case(emu_backend)


def test_mse(emu_backend):
    oc.mse_case(emu_backend)


def test_avgpool(emu_backend):
    oc.avgpool_case(emu_backend)
    oc.avgpool_case(emu_backend, 37, 41)


def test_lstm_gates(emu_backend):
    oc.lstm_case(emu_backend)


def test_pck_accuracy(emu_backend, golden_dir):
    oc.accuracy_case(emu_backend, golden_dir)


def test_targets(emu_backend, golden_dir):
    oc.targets_case(emu_backend, golden_dir)
    oc.normalize_case(emu_backend)


def test_argmax(emu_backend, golden_dir):
    oc.argmax_case(emu_backend, golden_dir)


@pytest.mark.parametrize("math,tol", [("bf16x3", 2e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("cfg", [
    (2, 64, 6, 5, 32, 1, 1, 0, 1),        # 1x1
    (1, 64, 7, 7, 72, 3, 1, 1, 1),        # 3x3, ragged N tile
    (1, 64, 7, 7, 64, 3, 1, 3, 3),        # dilated, taps in the padding
    (2, 64, 9, 9, 64, 3, 2, 1, 1),        # stride 2 (data gradient through the parity gather, MODE 1)
    (1, 128, 5, 5, 64, 1, 1, 0, 1),       # two K slices per tap
])
def test_conv_bf16_operand_kernels(emu_backend, cfg, math, tol):
    """split-bf16 (fp32-equivalent) and plain bf16 MFMA kernels: forward + data gradient (weight gradient stays fp32)."""
    from unipose_amd import ops
    n, c, h, w, k, r, s, p, d = cfg
    ops.set_conv_math(math)
    try:
        errs = oc.conv_case(emu_backend, n, c, h, w, k, r, s, p, d, tol=tol)
    finally:
        ops.set_conv_math("f32")
    if math == "bf16":
        assert errs["y"] > 1e-4       # the bf16 kernel really ran (an fp32 result would sit at 1e-6)


@pytest.fixture
def guard_pages(emu_backend, monkeypatch):
    """Every tensor the host layer allocates (outputs, packed weights, partial slabs, gradients) ends right before a
    PROT_NONE page and float outputs start as NaN: an out-of-bounds access by a kernel kills the test, an element
    it forgot to write poisons the comparison."""
    import sys
    sys.path.insert(0, __file__.rsplit("/", 1)[0] + "/emu")
    import guard_alloc
    from unipose_amd import ops
    monkeypatch.setattr(ops, "torch", guard_alloc.TorchProxy())
    monkeypatch.setattr(ops, "_WS", {})
    monkeypatch.setattr(ops, "_PACK_CACHE", {})
    monkeypatch.setattr(ops, "_PACK16_CACHE", {})
    return guard_alloc


@pytest.mark.parametrize("cfg", [
    (2, 16, 6, 5, 32, 1, 1, 0, 1, False, False),
    (2, 16, 9, 9, 24, 3, 2, 1, 1, False, False),
    (1, 3, 20, 18, 8, 7, 2, 3, 1, False, False),      # generic path, ragged K slice
    (2, 32, 5, 5, 14, 1, 1, 0, 1, True, False),       # K=14 -> Kp=16 data gradient (the bug that faulted on the GPU)
    (1, 15, 6, 6, 14, 3, 1, 1, 1, True, True),
    (1, 128, 5, 5, 32, 3, 1, 1, 1, False, False),     # double-buffered loop
    (1, 64, 6, 6, 17, 1, 1, 0, 1, True, False),       # K=17 -> ldy 20
    (2, 96, 6, 6, 72, 3, 1, 1, 1, False, False),      # K-split tail tiles (13 uneven parts)
    (1, 15, 8, 7, 14, 3, 2, 1, 1, True, False),       # stride-2 parity classes, generic path
    (2, 32, 7, 7, 16, 1, 2, 0, 1, False, False),      # 1x1 stride 2: memset + one class
    (1, 32, 5, 5, 16, 3, 1, 6, 6, False, False),      # tap skipping down to one tap + empty K-split parts
])
def test_conv_no_out_of_bounds(guard_pages, cfg):
    n, c, h, w, k, r, s, p, d, bias, relu = cfg
    oc.conv_case(torch.device("cpu"), n, c, h, w, k, r, s, p, d, bias=bias, relu=relu)


@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
def test_conv_bf16_no_out_of_bounds(guard_pages, math):
    from unipose_amd import ops
    ops.set_conv_math(math)
    try:
        oc.conv_case(torch.device("cpu"), 1, 64, 7, 7, 72, 3, 1, 1, 1, tol=3e-2)
        oc.conv_case(torch.device("cpu"), 2, 64, 9, 9, 64, 3, 2, 1, 1, tol=3e-2)
    finally:
        ops.set_conv_math("f32")


WGRAD_BF16_CASES = [
    # n, c,  h,  w,  k, r, stride, pad, dil, bias
    (2, 16, 9, 9, 24, 3, 2, 1, 1, False),        # stride 2, 64x64 tile (K <= 64), rectangles in output coordinates
    (1, 3, 20, 18, 8, 7, 2, 3, 1, False),        # stem 7x7 stride 2: Cp = 4, many taps per column tile
    (1, 15, 12, 12, 8, 11, 1, 5, 1, True),       # 11x11 (Cp = 16), bias gradient
    (2, 32, 8, 8, 20, 1, 2, 0, 1, False),        # 1x1 stride 2: no rectangle table (plain reduction domain)
    (1, 32, 4, 4, 16, 3, 1, 5, 5, False),        # dilation > H: eight taps never live (empty rectangles)
    (3, 32, 23, 23, 72, 3, 1, 18, 18, False),    # WASP geometry, three images, 128-row tile with K = 72 (ragged rows)
    (2, 160, 7, 7, 136, 3, 1, 1, 1, False),      # two row tiles x several column tiles, channel counts not multiples of 128
    (1, 64, 33, 35, 64, 1, 1, 0, 1, False),      # 1155 pixels: several slices + a ragged last one, split over workgroups
]


@pytest.mark.parametrize("cfg", WGRAD_BF16_CASES)
def test_wgrad_bf16_kernel(emu_backend, cfg):
    """Weight gradient on the bf16 MFMA (up_conv2d_bwd_weight_bf16: operands rounded to bf16 while staged, transposed by the
    pack itself, fp32 accumulation): every geometry class of the fp32 kernel's tests, with and without live rectangles."""
    from unipose_amd import _C, ops
    n, c, h, w, k, r, s, p, d, bias = cfg
    ops.set_conv_math("bf16")
    ops.WGRAD_BF16_ANY_WIDTH = True      # (the model keeps layers that are not 32-channel aligned on the fp32 MFMA)
    try:
        for rect in (1, 0):
            _C.check(_C.lib().up_conv_tune(b"wgrad_rect", rect), "tune")
            errs = oc.conv_case(emu_backend, n, c, h, w, k, r, s, p, d, bias=bias, tol=3e-2)
            assert 1e-5 < errs["dw"] < 2e-2, errs      # bf16 rounding is visible, yet far inside the tolerance
    finally:
        _C.lib().up_conv_tune(b"wgrad_rect", 1)
        ops.set_conv_math("f32")
        ops.WGRAD_BF16_ANY_WIDTH = False


def test_wgrad_bf16_no_out_of_bounds(guard_pages):
    from unipose_amd import ops
    ops.set_conv_math("bf16")
    ops.WGRAD_BF16_ANY_WIDTH = True
    try:
        for n, c, h, w, k, r, s, p, d, bias in WGRAD_BF16_CASES[:6]:
            oc.conv_case(torch.device("cpu"), n, c, h, w, k, r, s, p, d, bias=bias, tol=3e-2)
    finally:
        ops.set_conv_math("f32")
        ops.WGRAD_BF16_ANY_WIDTH = False


def test_conv_bn_no_out_of_bounds(guard_pages):
    oc.conv_bn_case(torch.device("cpu"), 3, 8, 9, 9, 72, 3, 2, 1, 1, relu=True, residual=True, train=True)


def test_small_ops_no_out_of_bounds(guard_pages, golden_dir):
    dev = torch.device("cpu")
    oc.layout_case(dev)
    oc.maxpool_case(dev, 1, 4, 7, 9)
    oc.bilinear_case(dev, 1, 4, 3, 5, 7, 9)
    oc.gap_case(dev, 2, 8, 3, 5)
    oc.concat_case(dev)
    oc.mse_case(dev)
    oc.avgpool_case(dev, 37, 41)
    oc.lstm_case(dev)
    oc.argmax_case(dev, golden_dir)


@pytest.mark.parametrize("per_cu", [2, 4])
def test_conv_all_tiles_split(emu_backend, per_cu):
    """Launches with fewer than two tiles per CU split EVERY tile along K (knob "split_per_cu"): forward, data gradient
    (incl. the addend that joins after the merge) and BatchNorm statistics from the merged accumulators."""
    from unipose_amd import _C
    lib = _C.lib()
    try:
        _C.check(lib.up_conv_tune(b"split_per_cu", per_cu), "split_per_cu")
        oc.conv_case(emu_backend, 2, 256, 9, 9, 80, 3, 1, 1, 1, bias=True, relu=True)        # 3 row tiles x 2 column tiles, 72 slices
        oc.conv_case(emu_backend, 1, 512, 7, 7, 64, 1, 1, 0, 1)                              # 1x1, 16 slices
        oc.conv_case(emu_backend, 3, 64, 23, 23, 32, 3, 1, 6, 6)                             # tap-sorted rows + split
        oc.dgrad_add_case(emu_backend, 1, 128, 6, 6, 64, 3, 1, 1, 1)
        oc.conv_bn_case(emu_backend, 2, 128, 5, 5, 48, 3, 1, 1, 1, relu=True, residual=True, train=True)
    finally:
        lib.up_conv_tune(b"split_per_cu", 2)


@pytest.mark.parametrize("cfg", [
    (3, 32, 5, 7, 32, True, True, "f32"),       # 8 channel groups of 4: the smallest row-strided geometry
    (2, 32, 9, 9, 128, True, False, "f32"),
    (5, 32, 3, 3, 64, False, True, "f32"),      # more row lanes than rows
    (3, 32, 5, 7, 64, True, True, "bf16"),      # 8 channel groups of 8
    (2, 32, 6, 5, 128, True, False, "bf16"),
])
def test_bn_row_strided_passes_match_flat_passes(emu_backend, cfg):
    import torch
    n, c, h, w, k, relu, residual, dt = cfg
    oc.bn_rows_ab_case(emu_backend, n, c, h, w, k, relu=relu, residual=residual,
                       dtype=torch.float32 if dt == "f32" else torch.bfloat16)


@pytest.mark.parametrize("cfg", [
    (3, 2, 32, 5, 7, 32, 1, True, False, "f32"),      # three groups of two images
    (5, 2, 32, 6, 6, 48, 3, True, True, "f32"),       # 3x3, residual, channel groups not a power of two (flat kernels)
    (2, 3, 32, 9, 9, 64, 1, False, True, "f32"),      # no ReLU: no bit array
    (3, 2, 32, 5, 8, 64, 1, True, True, "bf16"),
])
def test_grouped_batchnorm_matches_separate_calls(emu_backend, cfg):
    """ops.bn_groups(G): one convolution over G*n images, BatchNorm statistics / running updates / gradients of G calls"""
    import torch
    groups, n, c, h, w, k, r, relu, residual, dt = cfg
    e = oc.bn_groups_case(emu_backend, groups, n, c, h, w, k, r=r, relu=relu, residual=residual,
                          dtype=torch.float32 if dt == "f32" else torch.bfloat16)
    assert not e["grouped_tiles"]        # <= 256 rows per group: float64 statistics from y, plain tiling


@pytest.mark.parametrize("cfg", [
    (3, 2, 32, 13, 11, 32, 1, True, False),      # 286 rows per group = 4 tiles of 64 + 30 rows: a plain tiling would straddle
    (2, 3, 32, 10, 10, 64, 3, True, True),       # 3x3: tap-sorted rows inside every group, residual
    (5, 1, 64, 17, 16, 128, 1, False, True),     # five groups of ONE image (272 rows), no ReLU, two K slices
    (2, 2, 32, 12, 12, 136, 3, True, False),     # N = 136: ragged column tile; 288 rows per group
])
def test_grouped_batchnorm_tiles_per_group(emu_backend, cfg):
    """more than 256 rows per group in fp32: every group is tiled on its own by the direct-to-LDS kernel (up_conv2d_fwd_grouped), the
    epilogue's Welford partials are per group and the extra statistics pass over y is gone — same checks against G torch calls"""
    groups, n, c, h, w, k, r, relu, residual = cfg
    e = oc.bn_groups_case(emu_backend, groups, n, c, h, w, k, r=r, relu=relu, residual=residual, grouped_fwd=True)
    assert e["grouped_tiles"]


@pytest.mark.parametrize("groups,rows,c", [(3, 40000, 8), (5, 17 * 256 - 7, 12), (8, 300, 4)])
def test_grouped_statistics_and_finalize_many_tiles(emu_backend, groups, rows, c):
    oc.bn_group_stats_case(emu_backend, groups, rows, c)


@pytest.mark.parametrize("cfg", [
    (3, 2, 32, 13, 11, 32, 32, 1),     # 286 rows per group: 4 tiles of 64 + 30 rows
    (2, 3, 32, 10, 10, 64, 32, 3),     # the consumer is a 3x3 (tap-sorted rows inside every group)
    (5, 1, 32, 17, 16, 32, 128, 1),    # five groups of one image
    (2, 1, 32, 9, 9, 32, 32, 1),       # 81 rows per group: float64 statistics in the forward, fused reduction in the backward
])
def test_grouped_fused_reduction(emu_backend, cfg):
    print(oc.bn_groups_chain_case(emu_backend, *cfg))


def test_bn_large_mean_is_applied_centred(emu_backend):
    print(oc.bn_large_mean_case(emu_backend))


def test_bn_small_batch_statistics_are_exact(emu_backend):
    oc.bn_small_batch_case(emu_backend)
    oc.bn_small_batch_case(emu_backend, n=7, c=64, k=40, seed=31)
    oc.conv_bn_case(emu_backend, 4, 64, 1, 1, 32, 1, 1, 0, 1, relu=True, train=True)      # gradients through the same path


@pytest.mark.parametrize("math", ["f32", "bf16"])
@pytest.mark.parametrize("fused", [True, False])
def test_optimizer_step_makes_the_packed_weights_stale(emu_backend, math, fused):
    oc.optimizer_stale_case(emu_backend, math, fused)


def test_batched_repack_equals_single_pack(emu_backend):
    """ONE up_pack_weights_batched / up_pack_weights_bf16_batched launch over many parameters against up_pack_weights /
    up_pack_weights_bf16 per parameter, bit for bit: padded input and output channels, 7x7 / 3x3 / 1x1, the parity-class-major
    data-gradient image of stride-2 convolutions, rows shorter and longer than a wavefront.  The bf16 lo planes are only kept
    current for the split-bf16 arithmetic (the one that reads them); switching to it re-packs them."""
    import ctypes
    from unipose_amd import _C, ops
    gen = torch.Generator().manual_seed(5)
    geo = [(64, 3, 7, 2, 3, 1), (32, 32, 3, 1, 1, 1), (96, 64, 3, 2, 1, 1), (17, 64, 1, 1, 0, 1), (128, 32, 1, 2, 0, 1),
           (40, 96, 3, 1, 2, 2), (14, 15, 3, 1, 1, 1), (200, 72, 1, 1, 0, 1)]
    ws, descs = [], []
    for k, c, r, stride, pad, dil in geo:
        w = torch.nn.Parameter(torch.randn(k, c, r, r, generator=gen))
        x = torch.zeros(1, 12, 12, ops.rup32(c) if c % 32 == 0 else ops.rup4(c))
        ws.append(w)
        descs.append(ops.make_desc(x, w, ops.ConvCfg(stride, pad, dil)))
    L = _C.lib()
    try:
        for math in ("f32", "bf16", "bf16x3", "bf16"):
            bf16 = math != "f32"
            ops.set_conv_math(math)
            use = [(w, d) for w, d in zip(ws, descs) if not bf16 or (d.Cp % 32 == 0)]
            get = ops._packed_bf16 if bf16 else ops._packed
            for w, d in use:
                get(w, d)                                            # registers the parameter (single-pack launches)
            with torch.no_grad():
                for w, _ in use:
                    w.mul_(1.5).add_(0.25)                           # new values, new version
            first = get(*use[0])                                     # the first stale hit re-packs EVERY registered parameter
            for w, d in use:
                wf, wd = get(w, d)
                rf, rd = torch.empty_like(wf), torch.empty_like(wd)
                if bf16:
                    _C.check(L.up_pack_weights_bf16(ctypes.byref(d), w.data_ptr(), rf[0].data_ptr(), rf[1].data_ptr(), rd[0].data_ptr(),
                                                    rd[1].data_ptr(), 0), "pack_weights_bf16")
                    if math != "bf16x3":                             # lo planes: not maintained, not read
                        wf, wd, rf, rd = wf[0], wd[0], rf[0], rd[0]
                else:
                    _C.check(L.up_pack_weights(ctypes.byref(d), w.data_ptr(), rf.data_ptr(), rd.data_ptr(), 0), "pack_weights")
                assert torch.equal(wf, rf), ("forward image", math, tuple(w.shape))
                assert torch.equal(wd, rd), ("data-gradient image", math, tuple(w.shape), d.stride)
            assert first[0].data_ptr() == get(*use[0])[0].data_ptr()     # persistent buffers
    finally:
        ops.set_conv_math("f32")


@pytest.mark.parametrize("cfg", [
    dict(n=2, c=32, h=13, w=11, k1=64, k2=32),                    # 286 rows: 5 tiles, one group, ragged last tile
    dict(n=2, c=32, h=48, w=47, k1=96, k2=160, r2=1),             # 4512 rows: 71 tiles = 3 groups; 96 / 160 channels: ragged columns
    dict(n=3, c=32, h=60, w=50, k1=32, k2=32, r2=3),              # 9000 rows: 141 tiles = 5 groups (level 2 uses every row lane); 3x3 consumer
    dict(n=2, c=32, h=20, w=20, k1=64, k2=64, cus=6),             # shrunk chip: K-split tail tiles, only the finisher arrives
    dict(n=1, c=20, h=20, w=19, k1=32, k2=32),                    # 20 input channels: the register-staged generic kernel carries the ticket
])
def test_bn_finalize_folded_into_the_producing_launch(emu_backend, cfg):
    print(oc.bn_fold_case(emu_backend, **cfg))


def test_bn_finalize_folded_bf16_storage(emu_backend):
    from unipose_amd import ops
    ops.set_conv_math("bf16s")
    try:
        print(oc.bn_fold_case(emu_backend, n=2, c=32, h=30, w=33, k1=64, k2=96, dtype=torch.bfloat16))
    finally:
        ops.set_conv_math("f32")


def test_debug_pack_mode_catches_a_weight_edited_behind_the_cache(emu_backend):
    """UNIPOSE_DEBUG_PACK: a weight changed through .data (no version bump, no optimizer hook) makes the next convolution raise
    instead of running on the stale packed image; ops.invalidate_packed_weights() is the cure."""
    from unipose_amd import _C, ops
    w = torch.nn.Parameter(torch.randn(32, 32, 3, 3) * 0.05)
    x = torch.randn(1, 6, 6, 32)
    cfg = ops.ConvCfg(1, 1, 1)
    prev = ops.DEBUG_PACK
    ops.DEBUG_PACK = True
    try:
        y0 = ops.conv_fwd_raw(x, w, cfg)[0].clone()
        assert torch.equal(ops.conv_fwd_raw(x, w, cfg)[0], y0)                 # an untouched weight passes the check
        w.data.mul_(2.0)                                                     # behind the cache's back
        with pytest.raises(_C.UniPoseHipError, match="stale packed weight image"):
            ops.conv_fwd_raw(x, w, cfg)
        ops.invalidate_packed_weights()
        assert torch.allclose(ops.conv_fwd_raw(x, w, cfg)[0], 2.0 * y0, rtol=1e-6, atol=1e-6)
    finally:
        ops.DEBUG_PACK = prev


@pytest.mark.parametrize("cfg", [
    (3, 2, 32, 13, 11, 32, 32, 1),     # 286 rows per group: 2 statistics chunks / 5 data-gradient tiles per group
    (5, 1, 32, 40, 41, 64, 96, 1),     # five groups, 1640 rows each: 7 chunks, 26 tiles per group
    (2, 3, 32, 24, 30, 32, 32, 3),     # 2160 rows per group = 34 tiles: two level-1 rows per group; 3x3 consumer
])
def test_grouped_bn_finalize_folded(emu_backend, cfg):
    print(oc.bn_groups_fold_case(emu_backend, *cfg))


def _shared_weight_graph(dev, uses, seed=3):
    """`uses` convolutions (with bias) of a NON-LEAF weight built by torch.cat from two leaves, like the ConvLSTM cell's stacked gates"""
    import torch
    from unipose_amd import ops
    g = torch.Generator().manual_seed(seed)
    wa = (torch.randn(8, 16, 3, 3, generator=g) * 0.1).to(dev).requires_grad_(True)
    wb = (torch.randn(8, 16, 3, 3, generator=g) * 0.1).to(dev).requires_grad_(True)
    ba = torch.randn(16, generator=g).to(dev).requires_grad_(True)
    xs = [torch.randn(2, 5, 5, 16, generator=g).to(dev) for _ in range(uses)]
    w, b = torch.cat([wa, wb], 0), ba * 1.0
    ys = [ops.ConvBias.apply(x, w, b, ops.ConvCfg(1, 1, 1), False) for x in xs]
    return (wa, wb, ba), w, ys


def test_deferred_wgrad_hands_a_shared_non_leaf_weight_one_summed_gradient(emu_backend):
    """ops.deferred_wgrad, non-leaf weight used by three convolutions: autograd receives ONE gradient for it (the last use hands over
    the sum the library accumulated) and the leaves behind the cat get what the engine's own accumulation gives them."""
    import torch
    from unipose_amd import ops
    leaves, w, ys = _shared_weight_graph(emu_backend, 3)
    seen = []
    w.register_hook(lambda g: seen.append(1))
    sum((y * y).sum() for y in ys).backward()
    ref = [p.grad.clone() for p in leaves]
    assert len(seen) == 1                       # (hooks on a tensor fire once, on the engine's accumulated gradient)
    leaves2, w2, ys2 = _shared_weight_graph(emu_backend, 3)
    calls = []
    orig = ops.conv_bwd_weight_raw

    def counting(*a, **k):
        calls.append(bool(k.get("accumulate", k.get("out") is not None)))
        return orig(*a, **k)
    ops.conv_bwd_weight_raw = counting
    try:
        with ops.deferred_wgrad():
            sum((y * y).sum() for y in ys2).backward()
    finally:
        ops.conv_bwd_weight_raw = orig
    assert calls == [False, True, True]         # first use writes the buffer, the later ones add to it inside the reduce pass
    for a, b in zip(ref, [p.grad for p in leaves2]):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


def test_deferred_wgrad_reports_a_shared_use_without_gradient(emu_backend):
    """a forward use of a shared non-leaf weight whose gradient never arrives inside the context must not swallow the others' sum"""
    import pytest
    from unipose_amd import ops
    leaves, w, ys = _shared_weight_graph(emu_backend, 3)
    with pytest.raises(RuntimeError, match="never arrived"):
        with ops.deferred_wgrad():
            sum((y * y).sum() for y in ys[:2]).backward()          # the third use is left out of the loss


def test_stem_kernel_matches_generic_kernel_emu(emu_backend):
    oc.stem_ab_case(emu_backend, 1, 256)            # one image, 128 x 128 output pixels: every tile is one run of a row
    oc.stem_ab_case(emu_backend, 2, 264, seed=3)    # 132 x 132: tiles of two runs, across the image boundary, ragged last tile
