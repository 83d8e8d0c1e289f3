"""Per-operator parity on a real MI355X at the reference network's own layer shapes,
through the C ABI (ctypes -> libunipose_hip.so).  Checker: plain torch fp32 on the CPU."""
import pytest
import torch

import op_cases as oc

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# n, c, h, w, k, r, stride, pad, dil, bias, relu           (one row per distinct conv flavour of T1)
CONVS = [
    (2, 1024, 23, 23, 256, 1, 1, 0, 1, False, False),      # layer3 1x1 reduce            (K1)
    (2, 256, 23, 23, 1024, 1, 1, 0, 1, False, False),      # layer3 1x1 expand            (K1)
    (2, 256, 23, 23, 256, 3, 1, 1, 1, False, False),       # layer3 3x3                   (K3)
    (2, 512, 23, 23, 512, 3, 1, 2, 2, False, False),       # layer4 dilated d2            (K5)
    (2, 512, 23, 23, 512, 3, 1, 8, 8, False, False),       # layer4 dilated d8            (K5)
    (2, 256, 23, 23, 256, 3, 1, 18, 18, False, False),     # WASP d18: 77% of taps in padding (K5)
    (2, 256, 23, 23, 256, 3, 1, 12, 12, False, False),     # WASP d12
    (2, 256, 23, 23, 256, 3, 1, 6, 6, False, False),       # WASP d6
    (1, 256, 46, 46, 256, 3, 1, 24, 24, False, False),     # WASP at 736 input: dil 24 on 46x46 (K5)
    (2, 3, 368, 368, 64, 7, 2, 3, 1, False, False),        # stem 7x7 s2                  (K6)
    (2, 128, 92, 92, 128, 3, 2, 1, 1, False, False),       # layer2.0 3x3 s2              (K4)
    (2, 256, 92, 92, 512, 1, 2, 0, 1, False, False),       # layer2.0 downsample 1x1 s2   (K2)
    (2, 64, 92, 92, 64, 3, 1, 1, 1, False, False),         # layer1 3x3
    (2, 304, 46, 46, 256, 3, 1, 1, 1, False, False),       # decoder 3x3 304->256
    (2, 256, 46, 46, 17, 1, 1, 0, 1, True, False),         # decoder head 1x1 + bias, K=17
    (2, 256, 92, 92, 48, 1, 1, 0, 1, False, False),        # decoder low-level 1x1 -> 48
    (2, 2048, 1, 1, 256, 1, 1, 0, 1, False, True),         # WASP GAP branch (M = batch)
    (1, 15, 46, 46, 128, 11, 1, 5, 1, True, True),         # LSTM head 11x11 15->128      (K17)
    (1, 128, 46, 46, 128, 11, 1, 5, 1, True, True),        # LSTM head 11x11 128->128     (K17)
    (2, 32, 46, 46, 60, 3, 1, 1, 1, True, False),          # fused ConvLSTM gates over cat(x,h)
    (2, 128, 46, 46, 14, 1, 1, 0, 1, True, True),          # LSTM head conv5
]


@pytest.mark.parametrize("cfg", CONVS, ids=lambda c: "c%d_%dx%d_k%d_r%d_s%d_d%d" % (c[1], c[2], c[3], c[4], c[5], c[6], c[8]))
def test_conv_fwd_bwd(cfg):
    n, c, h, w, k, r, s, p, d, bias, relu = cfg
    oc.conv_case(DEV, n, c, h, w, k, r, s, p, d, bias=bias, relu=relu, tol=5e-5)


CONV_BN = [
    # n, c, h, w, k, r, stride, pad, dil, relu, residual, train
    (2, 64, 92, 92, 256, 1, 1, 0, 1, True, True, True),     # layer1 expand + residual + relu
    (2, 512, 46, 46, 128, 1, 1, 0, 1, True, False, True),
    (2, 256, 23, 23, 256, 3, 1, 1, 1, True, False, True),
    (2, 256, 23, 23, 256, 3, 1, 12, 12, True, False, True),  # _AtrousModule
    (2, 512, 46, 46, 1024, 1, 2, 0, 1, False, False, True),  # downsample: no relu
    (2, 3, 184, 184, 64, 7, 2, 3, 1, True, False, True),     # stem
    (4, 2048, 1, 1, 256, 1, 1, 0, 1, True, False, True),     # GAP branch BN over batch only
    (2, 1280, 23, 23, 256, 1, 1, 0, 1, True, False, True),   # WASP fuse conv1
    (2, 256, 23, 23, 1024, 1, 1, 0, 1, True, True, False),   # eval statistics with grad
]


@pytest.mark.parametrize("cfg", CONV_BN, ids=lambda c: "c%d_%dx%d_k%d_r%d_s%d_d%d_%s" % (c[1], c[2], c[3], c[4], c[5], c[6], c[8], "train" if c[11] else "eval"))
def test_conv_bn_act(cfg):
    n, c, h, w, k, r, s, p, d, relu, res, train = cfg
    oc.conv_bn_case(DEV, n, c, h, w, k, r, s, p, d, relu=relu, residual=res, train=train, tol=1e-4)


def test_layout():
    oc.layout_case(DEV)


def test_maxpool():
    oc.maxpool_case(DEV, 2, 64, 184, 184)        # stem pool
    oc.maxpool_case(DEV, 2, 48, 92, 92)          # decoder pool
    oc.maxpool_case(DEV, 1, 8, 9, 10)


@pytest.mark.parametrize("shape", [(2, 256, 23, 23, 46, 46), (2, 256, 1, 1, 23, 23), (1, 20, 46, 46, 368, 368),
                                   (1, 256, 46, 46, 92, 92)])
def test_bilinear(shape):
    oc.bilinear_case(DEV, *shape)


def test_gap():
    oc.gap_case(DEV, 2, 2048, 23, 23)


def test_concat():
    oc.concat_case(DEV)


def test_dropout():
    oc.dropout_case(DEV)


def test_mse():
    oc.mse_case(DEV)


def test_avgpool():
    oc.avgpool_case(DEV, 368, 368)
    oc.avgpool_case(DEV, 37, 41)


def test_lstm_gates():
    oc.lstm_case(DEV)


def test_argmax(golden_dir):
    oc.argmax_case(DEV, golden_dir)


def test_argmax_full_size_properties():
    """B=32 x 17 maps of 92x92 (BASELINE config 5 output size): bit-exact vs torch's first-max argmax."""
    from unipose_amd import ops
    g = torch.Generator().manual_seed(3)
    hm = torch.randn(32, 17, 92, 92, generator=g)
    hm[3, 4] = hm[3, 4].round()                      # plenty of ties
    preds, mx, idx = ops.heatmap_argmax(hm.to(DEV))
    flat = hm.reshape(32, 17, -1)
    ref = flat.numpy().argmax(2)
    assert (idx.cpu().numpy() == ref).all()
    assert torch.equal(mx.cpu()[..., 0], flat.max(2).values)


BF16_CONVS = [
    (2, 1024, 23, 23, 256, 1, 1, 0, 1),
    (2, 256, 23, 23, 1024, 1, 1, 0, 1),
    (2, 256, 23, 23, 256, 3, 1, 1, 1),
    (2, 512, 23, 23, 512, 3, 1, 4, 4),
    (2, 256, 23, 23, 256, 3, 1, 18, 18),
    (2, 128, 92, 92, 128, 3, 2, 1, 1),
    (2, 256, 92, 92, 512, 1, 2, 0, 1),
    (2, 64, 92, 92, 64, 3, 1, 1, 1),
    (2, 320, 46, 46, 256, 3, 1, 1, 1),
]


@pytest.mark.parametrize("math,tol", [("bf16x3", 2e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("cfg", BF16_CONVS, ids=lambda c: "c%d_%dx%d_k%d_r%d_s%d_d%d" % (c[1], c[2], c[3], c[4], c[5], c[6], c[8]))
def test_conv_bf16_operand_kernels(cfg, math, tol):
    from unipose_amd import ops
    n, c, h, w, k, r, s, p, d = cfg
    ops.set_conv_math(math)
    try:
        errs = oc.conv_case(DEV, n, c, h, w, k, r, s, p, d, tol=tol)
    finally:
        ops.set_conv_math("f32")
    if math == "bf16":
        assert errs["y"] > 1e-4
    else:
        assert errs["y"] < 5e-5, errs          # split-bf16 keeps ~16 mantissa bits per operand


@pytest.mark.parametrize("cfg", [
    (32, 1024, 23, 23, 256, 1, 1, 0, 1),     # layer3 conv1 at the benchmark batch: 1064-tile launch with K-split tail
    (2, 256, 92, 92, 64, 1, 1, 0, 1),        # layer1 conv1
    (8, 256, 23, 23, 256, 3, 1, 1, 1),
])
def test_dgrad_with_addend(cfg):
    oc.dgrad_add_case(DEV, *cfg)


def test_pck_accuracy(golden_dir):
    oc.accuracy_case(DEV, golden_dir)


def test_targets(golden_dir):
    oc.targets_case(DEV, golden_dir)
    oc.normalize_case(DEV)


@pytest.mark.parametrize("cfg", [
    (4, 64, 23, 23, 256, True, False, "f32"),
    (4, 64, 23, 23, 1024, True, True, "f32"),      # 256 channel groups: one lane per group
    (2, 64, 23, 23, 2048, False, True, "f32"),     # 512 groups: two workgroup columns
    (2, 64, 92, 92, 64, True, False, "f32"),
    (4, 64, 46, 46, 1024, True, True, "bf16"),
    (2, 64, 92, 92, 64, True, False, "bf16"),
    (2, 64, 23, 23, 2048, False, True, "bf16"),
])
def test_bn_row_strided_passes_match_flat_passes(cfg):
    n, c, h, w, k, relu, residual, dt = cfg
    oc.bn_rows_ab_case(DEV, n, c, h, w, k, relu=relu, residual=residual, dtype=torch.float32 if dt == "f32" else torch.bfloat16)


@pytest.mark.parametrize("cfg", [
    (5, 8, 64, 23, 23, 256, 1, True, False, "f32"),      # configs[3] geometry: five frames of eight images
    (5, 8, 64, 23, 23, 1024, 1, True, True, "f32"),
    (5, 2, 64, 46, 46, 64, 3, True, False, "f32"),
    (3, 2, 64, 23, 23, 48, 1, True, True, "f32"),        # channel groups not a power of two
    (5, 4, 64, 23, 23, 256, 1, True, True, "bf16"),
])
def test_grouped_batchnorm_matches_separate_calls(cfg):
    groups, n, c, h, w, k, r, relu, residual, dt = cfg
    for fwd in (False, True):
        e = oc.bn_groups_case(DEV, groups, n, c, h, w, k, r=r, relu=relu, residual=residual,
                              dtype=torch.float32 if dt == "f32" else torch.bfloat16, grouped_fwd=fwd)
        # fp32 with 4-aligned output channels: every group tiled on its own (4232 rows per group = 66 tiles of 64 + 8 rows)
        assert e["grouped_tiles"] == (fwd and dt == "f32" and k % 4 == 0)


@pytest.mark.parametrize("cfg", [
    (5, 8, 1024, 23, 23, 256, 256, 3),    # configs[3]: five frames of eight images, layer3 bn1 reduced by conv2's data gradient
    (5, 8, 256, 23, 23, 256, 1024, 1),    # ... bn2 by conv3's
    (5, 8, 64, 92, 92, 64, 64, 3),        # layer1 (67 712 rows per group, a multiple of the tile height)
    (3, 2, 64, 23, 23, 64, 128, 1),
])
def test_grouped_fused_reduction(cfg):
    print(oc.bn_groups_chain_case(DEV, *cfg, tol=1e-4))


@pytest.mark.parametrize("groups,rows,c", [(3, 40000, 8), (5, 4232, 256), (8, 67712, 64)])
def test_grouped_statistics_and_finalize_many_tiles(groups, rows, c):
    oc.bn_group_stats_case(DEV, groups, rows, c)


def test_bn_large_mean_is_applied_centred():
    print(oc.bn_large_mean_case(DEV))


def test_bn_small_batch_statistics_are_exact():
    oc.bn_small_batch_case(DEV)
    oc.bn_small_batch_case(DEV, n=7, c=64, k=40, seed=31)
    oc.bn_small_batch_case(DEV, n=32, c=2048, k=256, seed=33)                     # the GAP branch of the headline batch
    oc.conv_bn_case(DEV, 4, 64, 1, 1, 32, 1, 1, 0, 1, relu=True, train=True)


@pytest.mark.parametrize("math", ["f32", "bf16"])
@pytest.mark.parametrize("nweights", [3, 40])
def test_optimizer_step_makes_the_packed_weights_stale(math, nweights):
    """(40 parameters: the batched re-pack is split between the current and the side stream, both read orders)"""
    oc.optimizer_stale_case(DEV, math, True, nweights)


@pytest.mark.parametrize("cfg", [
    dict(n=32, c=1024, h=23, w=23, k1=256, k2=256, r2=3),         # layer3 bn1 (265 tiles = 9 groups), merged by conv2's data gradient
    dict(n=32, c=256, h=23, w=23, k1=256, k2=1024),               # bn2 by conv3's; 1024 channels = 16 ticket columns
    dict(n=8, c=64, h=92, w=92, k1=64, k2=256),                   # layer1: 1058 partial rows per column
    dict(n=4, c=32, h=46, w=47, k1=96, k2=160),                   # ragged tiles and columns
    dict(n=2, c=20, h=30, w=30, k1=32, k2=32),                    # register-staged generic kernel
])
def test_bn_finalize_folded_into_the_producing_launch(cfg):
    print(oc.bn_fold_case(DEV, **cfg))


def test_bn_finalize_folded_bf16_storage():
    from unipose_amd import ops
    ops.set_conv_math("bf16s")
    try:
        print(oc.bn_fold_case(DEV, n=16, c=256, h=46, w=46, k1=256, k2=1024, dtype=torch.bfloat16))
    finally:
        ops.set_conv_math("f32")


def test_bn_fold_is_deterministic_under_load():
    """The last arriver differs from launch to launch; the merged statistics must not: 20 repetitions of a 1060-tile launch with
    co-running work on a second stream, every result bit-equal to the first."""
    from unipose_amd import ops
    torch.manual_seed(5)
    conv, bn = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(DEV), torch.nn.BatchNorm2d(256).to(DEV).train()
    x = torch.randn(32, 23, 23, 256, device=DEV)
    noise = torch.randn(64 << 20, device=DEV)
    side = torch.cuda.Stream()
    ref = None
    for rep in range(20):
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0)
        with torch.cuda.stream(side):
            for _ in range(3):
                noise.mul_(1.0001)
        with torch.no_grad():
            y = ops.conv_bn_act(x, conv, bn, relu=True)
        got = (y.clone(), bn.running_mean.clone(), bn.running_var.clone())
        torch.cuda.synchronize()
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), rep


@pytest.mark.parametrize("cfg", [
    (5, 8, 1024, 23, 23, 256, 256, 3),    # configs[3]: five frames of eight images, layer3
    (5, 8, 64, 92, 92, 64, 64, 3),        # layer1: 67 712 rows per group
    (3, 2, 64, 23, 23, 64, 128, 1),
])
def test_grouped_bn_finalize_folded(cfg):
    print(oc.bn_groups_fold_case(DEV, *cfg))


@pytest.mark.gpu
@pytest.mark.parametrize("n,size", [(1, 256), (2, 264), (4, 368), (3, 736)])
def test_stem_kernel_matches_generic_kernel(n, size):
    oc.stem_ab_case(DEV, n, size, seed=size)
