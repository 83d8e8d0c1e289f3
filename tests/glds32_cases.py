"""A/B parity of the two exact-fp32 implicit-GEMM generations (shared by the emulator and the GPU tests).

`igemm_glds32_kernel` (unipose_amd/csrc/f32_glds.h: operands HBM -> LDS by LDS-DMA, LDS-transposed 16-byte-store epilogue)
must reproduce the register-staged `igemm_kernel<..., MODE 2>` EXACTLY: the LDS image, the fragment lane map, the k order
and the MFMA are the same, skipped taps only contribute exact zeros, the tail split cuts the same slices and merges in the
same order, the BatchNorm partials use the same arithmetic.  Outputs, statistics and data gradients are therefore compared
for equality (== on floats: a skipped tap may turn a -0 into +0); the register-staged kernel itself is pinned against torch in
op_cases.py.  The fused BatchNorm-backward reduction (BNRED) is compared with the separate reduce pass within fp32 round-off
of the sums (the two reduce in different orders)."""
import ctypes as C

import torch

from unipose_amd import _C, ops


def _g(seed):
    gen = torch.Generator()
    gen.manual_seed(seed)
    return gen


def _tune(**kw):
    for k, v in kw.items():
        _C.check(_C.lib().up_conv_tune(k.encode(), int(v)), k)


def _nhwc(t, dev, cp):
    n, c, h, w = t.shape
    y = torch.zeros(n, h, w, cp)
    y[..., :c] = t.permute(0, 2, 3, 1)
    return y.to(dev)


def _same(a, b, what):
    a, b = a.float().cpu(), b.float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not torch.equal(a, b):
        d = (a - b).abs()
        raise AssertionError(f"{what}: {int((d > 0).sum())} of {d.numel()} elements differ, max |diff| {float(d.max()):.3e}, "
                             f"first at {tuple(int(i) for i in (d > 0).nonzero()[0])}")


DEFAULTS = dict(glds32=1, glds32_epi=1, glds32_wgrad=1, tile_want=1500, cu_count=0, breg=0)


def conv_ab(dev, n, c, h, w, k, r, stride, pad, dil, *, tile_want, stats=False, affine=False, residual=False, relu=False,
            add=False, seed=0, cus=0, forms=(1, 0)):
    """forward (+ optional BatchNorm partials / folded epilogue / residual) and data gradient (+ optional addend) of one fp32
    convolution: both epilogue forms (`forms` = glds32_epi values) against glds32 = 0 under the same tile rule.
    `cus` shrinks the chip so that a small launch has whole rounds of tiles + K-split tail tiles."""
    cp, kp = ops.rup4(c), ops.rup4(k)
    x = _nhwc(torch.randn(n, c, h, w, generator=_g(seed)), dev, cp)
    wt = (torch.randn(k, c, r, r, generator=_g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5).to(dev)
    cfg = ops.ConvCfg(stride, pad, dil)
    kw = {}
    if affine:
        kw["scale"] = (0.5 + torch.rand(k, generator=_g(seed + 2))).to(dev)
        kw["shift"] = torch.randn(k, generator=_g(seed + 3)).to(dev)
        kw["bias"] = torch.randn(k, generator=_g(seed + 4)).to(dev)

    def run():
        d0 = ops.make_desc(x, wt, cfg)
        res = _nhwc(torch.randn(n, k, d0.P, d0.Q, generator=_g(seed + 5)), dev, kp) if residual else None
        y, d, st = ops.conv_fwd_raw(x, wt, cfg, residual=res, relu=relu, stats=stats, **kw)
        dy = _nhwc(torch.randn(n, k, d.P, d.Q, generator=_g(seed + 6)), dev, kp)
        addt = _nhwc(torch.randn(n, c, h, w, generator=_g(seed + 7)), dev, cp) if add else None
        dx = ops.conv_bwd_data_raw(dy, wt, d, x.shape, x.device, add=addt)
        dw, _ = ops.conv_bwd_weight_raw(x, dy, wt.shape, d, False)
        return y, st, dx, dw

    cnt = lambda name: int(_C.lib().up_conv_counter(name.encode()))
    try:
        _tune(tile_want=tile_want, cu_count=cus, glds32=0, glds32_wgrad=0)
        c0, w0 = cnt("glds32"), cnt("wgrad_glds32")
        y0, s0, dx0, dw0 = run()
        assert cnt("glds32") == c0 and cnt("wgrad_glds32") == w0, "glds32 = 0 still launched a direct-to-LDS kernel"
        for epi in forms:
            _tune(glds32=1, glds32_epi=epi, glds32_wgrad=1)
            c0, e0, w0, wide0 = cnt("glds32"), cnt("glds32_epi1"), cnt("wgrad_glds32"), cnt("glds32_wide")
            y1, s1, dx1, dw1 = run()
            assert cnt("glds32") > c0, "the case never reached the direct-to-LDS kernel"
            assert (cnt("glds32_wide") - wide0 == cnt("glds32") - c0) == (r * r > 32), "filters of more than 32 taps take the WIDE form, nothing else does"
            assert cnt("wgrad_glds32") > w0, "the weight gradient never reached the direct-to-LDS kernel"
            assert epi == 1 or cnt("glds32_epi1") == e0
            _same(dw1, dw0, "dw " + f"epi={epi}")
            tag = f"epi={epi}"
            _same(y1, y0, "y " + tag)
            if stats:
                _same(s1, s0, "BatchNorm partials " + tag)
            _same(dx1, dx0, "dx " + tag)
        if r == 1 and stride == 1:
            # round 6: the hybrid operand path (f32_glds.h BREG: weight fragments global -> registers, activations through LDS-DMA)
            # of the pointwise launches — same values, same k order: equal bits in every epilogue form
            for epi in forms:
                _tune(glds32=1, glds32_epi=epi, glds32_wgrad=1, breg=1)
                b0 = cnt("glds32_breg")
                y2, s2, dx2, _ = run()
                assert cnt("glds32_breg") > b0, "the pointwise case never reached the hybrid kernel"
                tag = f"breg, epi={epi}"
                _same(y2, y0, "y " + tag)
                if stats:
                    _same(s2, s0, "BatchNorm partials " + tag)
                _same(dx2, dx0, "dx " + tag)
    finally:
        _tune(**DEFAULTS)
    return y0


def bnred_case(dev, n, c, h, w, k, r, pad, dil, *, tile_want, add=False, mask_add=False, relu=True, seed=0, cus=0):
    """The data gradient of a convolution whose INPUT is z = relu(bn(y) (+ res)): the launch's epilogue reduces the
    BatchNorm-backward sums of that layer (up_conv2d_bwd_data_bnred) — against up_bn_bwd's own reduce pass on the same dz."""
    cp, kp = ops.rup4(c), ops.rup4(k)
    L = _C.lib()
    wt = (torch.randn(k, c, r, r, generator=_g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5).to(dev)
    cfg = ops.ConvCfg(1, pad, dil)
    x = _nhwc(torch.randn(n, c, h, w, generator=_g(seed)), dev, cp)       # stands for z: only its shape matters here
    d = ops.make_desc(x, wt, cfg)
    dy = _nhwc(torch.randn(n, k, d.P, d.Q, generator=_g(seed + 6)), dev, kp)
    addt = _nhwc(torch.randn(n, c, h, w, generator=_g(seed + 7)), dev, cp) if add else None
    ybn = _nhwc(torch.randn(n, c, h, w, generator=_g(seed + 8)) * 2 + 0.5, dev, cp)           # the producing layer's raw conv output
    rows = n * h * w
    bits = None
    if relu:
        zpos = (torch.rand(rows * c, generator=_g(seed + 9)) > 0.45)
        words = torch.zeros((rows * c + 31) // 32 * 32, dtype=torch.int64)
        words[:rows * c] = zpos.long()
        bits = (words.view(-1, 32) << torch.arange(32)).sum(1)
        bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32).to(dev)
    mean = torch.randn(c, generator=_g(seed + 10)).to(dev)
    invstd = (0.5 + torch.rand(c, generator=_g(seed + 11))).to(dev)
    gamma = (0.5 + torch.rand(c, generator=_g(seed + 12))).to(dev)
    abits = pre = None
    if add and mask_add:     # the addend is an UNMASKED gradient; its ReLU mask is applied by the epilogue
        apos = torch.rand(rows * c, generator=_g(seed + 13)) > 0.45
        aw = torch.zeros((rows * c + 31) // 32 * 32, dtype=torch.int64)
        aw[:rows * c] = apos.long()
        ab = (aw.view(-1, 32) << torch.arange(32)).sum(1)
        abits = torch.where(ab >= 2 ** 31, ab - 2 ** 32, ab).to(torch.int32).to(dev)
        pre = addt * apos.view(n, h, w, c).to(dev).float()
    try:
        _tune(tile_want=tile_want, cu_count=cus, glds32=1, glds32_epi=1)
        slot = ops.BnSlot(ybn, bits, mean, invstd, c)
        dx = ops.conv_bwd_data_raw(dy, wt, d, x.shape, x.device, add=addt, bn_slot=slot, add_bits=abits)
        assert slot.partial is not None, "the launch did not take the fused reduction"
        _tune(glds32=0)
        dx0 = ops.conv_bwd_data_raw(dy, wt, d, x.shape, x.device, add=addt if pre is None else pre)
    finally:
        _tune(**DEFAULTS)
    _same(dx, dx0, "dx (fused reduction on / off)")
    # the producing layer's backward from the pre-reduced sums against the three-pass form on the same dz
    outs = []
    for pre in (True, False):
        dyb = torch.empty_like(ybn)
        dres = torch.empty_like(ybn)
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        ws = ops.workspace(dx.device, L.up_bn_bwd_workspace(rows, c))
        if pre:
            _C.check(L.up_bn_bwd_prereduced_t(dx.data_ptr(), cp, _C_ptr(bits), ybn.data_ptr(), cp, gamma.data_ptr(), mean.data_ptr(),
                                              invstd.data_ptr(), int(relu), 1, dyb.data_ptr(), cp, dres.data_ptr(), cp,
                                              dgb[0].data_ptr(), dgb[1].data_ptr(), None, None, slot.partial.data_ptr(),
                                              slot.partial.shape[0], rows, c, 0, ops._stream(dx)), "bn_bwd_prereduced")
        else:
            _C.check(L.up_bn_bwd_acc_t(dx.data_ptr(), cp, None, 0, _C_ptr(bits), ybn.data_ptr(), cp, gamma.data_ptr(),
                                       mean.data_ptr(), invstd.data_ptr(), int(relu), 1, dyb.data_ptr(), cp, dres.data_ptr(), cp,
                                       dgb[0].data_ptr(), dgb[1].data_ptr(), None, None, ws.data_ptr(), ws.numel(), rows, c, 0,
                                       ops._stream(dx)), "bn_bwd")
        outs.append((dyb.cpu(), dres.cpu(), dgb.cpu()))
    (dy1, dr1, g1), (dy0, dr0, g0) = outs
    _same(dr1, dr0, "dres")
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 2e-5 * max(scale, 1.0), ("dgamma / dbeta", float((g1 - g0).abs().max()), scale)
    assert float((dy1 - dy0).abs().max()) <= 2e-5 * float(dy0.abs().max()), ("dy", float((dy1 - dy0).abs().max()))
    return dx


def _C_ptr(t):
    return None if t is None else t.data_ptr()


# (n, c, h, w, k, r, stride, pad, dil, tile_want, flags); channel counts are multiples of 32 (the kernel's eligibility)
SMALL = [
    dict(n=2, c=64, h=9, w=9, k=64, r=1, stride=1, pad=0, dil=1, tile_want=1, stats=True),           # 1x1, two slices, ragged row tile (162 rows)
    dict(n=3, c=64, h=7, w=7, k=128, r=3, stride=1, pad=1, dil=1, tile_want=1, stats=True),          # 128x128 tiles, 9 taps, tap-sorted
    dict(n=3, c=64, h=7, w=7, k=128, r=3, stride=1, pad=1, dil=1, tile_want=100000, stats=True),     # 64x64 tiles
    dict(n=3, c=32, h=7, w=7, k=136, r=3, stride=1, pad=1, dil=1, tile_want=3),                      # 64x128 / 128x64 by the rule, ragged N (136)
    dict(n=4, c=64, h=7, w=7, k=64, r=3, stride=1, pad=3, dil=3, tile_want=1, stats=True),           # dilated: dead taps, tap-sorted rows
    dict(n=1, c=32, h=23, w=23, k=64, r=3, stride=1, pad=18, dil=18, tile_want=100000, stats=True),  # WASP d = 18 geometry
    dict(n=2, c=96, h=6, w=6, k=72, r=3, stride=1, pad=1, dil=1, tile_want=1, add=True),             # three slices per tap, dgrad addend, N = 72
    dict(n=2, c=64, h=8, w=8, k=64, r=1, stride=1, pad=0, dil=1, tile_want=1, affine=True, relu=True, residual=True),   # folded eval epilogue
    dict(n=2, c=64, h=8, w=8, k=64, r=3, stride=1, pad=2, dil=2, tile_want=1, affine=True, relu=True),   # eval, no residual, tap-sorted
    dict(n=2, c=64, h=9, w=9, k=64, r=3, stride=2, pad=1, dil=1, tile_want=1, stats=True),           # stride 2: forward glds32, data gradient by parity classes
    dict(n=2, c=64, h=9, w=9, k=128, r=1, stride=2, pad=0, dil=1, tile_want=1, stats=True),          # 1x1 stride 2 (down-sampling branch)
    dict(n=1, c=160, h=5, w=5, k=34, r=1, stride=1, pad=0, dil=1, tile_want=1),                      # five slices, N = 34 (not a multiple of 4: dword epilogue)
    dict(n=1, c=32, h=6, w=6, k=32, r=5, stride=1, pad=2, dil=1, tile_want=1, stats=True),           # 25 taps (> 16: no tap sort, live-tap skipping only)
    dict(n=3, c=96, h=7, w=9, k=200, r=1, stride=1, pad=0, dil=1, tile_want=1, stats=True, add=True),  # 1x1, three slices (odd count: the unrolled loop's tail), N = 200 (ragged 128-wide tile), 189 rows
    dict(n=2, c=32, h=9, w=9, k=72, r=1, stride=1, pad=0, dil=1, tile_want=100000, stats=True),      # 1x1, ONE slice, 64x64 tiles, N = 72
]

# K-split tail tiles (chip shrunk to `cus` CUs): whole rounds + tail parts in one launch
SPLIT = [
    dict(n=3, c=64, h=7, w=7, k=128, r=3, stride=1, pad=1, dil=1, tile_want=100000, stats=True, cus=4),   # 64x64 tiles: 3 x 2 = 6 tiles = 4 + 2 tails x 2 parts
    dict(n=2, c=128, h=6, w=6, k=72, r=3, stride=1, pad=1, dil=1, tile_want=1, add=True, cus=0),          # one 128x128 tile: every tile split
    dict(n=4, c=64, h=7, w=7, k=64, r=3, stride=1, pad=3, dil=3, tile_want=100000, stats=True, cus=3),    # tap-sorted, tiles with different live taps
    dict(n=1, c=256, h=5, w=5, k=64, r=1, stride=1, pad=0, dil=1, tile_want=1, affine=True, relu=True, residual=True, cus=0),   # 1x1 8 slices, folded epilogue
    dict(n=4, c=160, h=7, w=7, k=128, r=1, stride=1, pad=0, dil=1, tile_want=100000, stats=True, cus=3),   # 1x1, 4 x 2 tiles of 64x64 on a 3-CU chip: K-split tails with an odd slice count per part
]

# more than 32 filter taps: the WIDE form of the forward / data-gradient kernel (separable row / column masks, every tap visited)
# against the register-staged per-slice-tap path.  32-aligned input AND output channels (the data gradient's input is dy).
WIDE = [
    dict(n=1, c=32, h=12, w=12, k=64, r=11, stride=1, pad=5, dil=1, tile_want=1),                          # the video head's 11x11: 121 taps
    dict(n=2, c=64, h=9, w=9, k=32, r=7, stride=1, pad=3, dil=1, tile_want=100000, add=True),              # 49 taps, two slices per tap, addend
    dict(n=1, c=32, h=10, w=10, k=128, r=11, stride=1, pad=5, dil=1, tile_want=1, affine=True, relu=True),   # folded epilogue (inference head)
    dict(n=2, c=32, h=8, w=8, k=32, r=7, stride=1, pad=6, dil=2, tile_want=1, stats=True),                  # dilated 7x7: dead rows / columns
    dict(n=2, c=32, h=9, w=9, k=128, r=7, stride=1, pad=3, dil=1, tile_want=100000, cus=4),                 # K-split tail tiles over 49 x 1 slices
]

# weight gradient only: shapes the convolution cases above do not reach (unaligned channel counts, many taps, stride 2, both
# stage forms: `cus` shrinks the chip so that the grid exceeds two workgroups per CU -> the one-stage form)
WGRAD = [
    dict(n=2, c=3, h=20, w=20, k=64, r=7, stride=2, pad=3, dil=1, cus=0),        # the stem: 3 -> 4 channels, 49 taps, stride 2
    dict(n=2, c=15, h=9, w=9, k=15, r=3, stride=1, pad=1, dil=1, cus=0),         # ConvLSTM gate: 15 -> 16 channels, K = 15
    dict(n=1, c=15, h=12, w=12, k=128, r=11, stride=1, pad=5, dil=1, cus=0),     # LSTM head 11x11
    dict(n=3, c=160, h=7, w=7, k=136, r=3, stride=1, pad=1, dil=1, cus=1),       # several tiles both ways, one-stage form (cus = 1)
    dict(n=3, c=64, h=23, w=23, k=64, r=3, stride=1, pad=18, dil=18, cus=2),     # live rectangles (WASP d = 18), one-stage form
    dict(n=4, c=64, h=10, w=10, k=256, r=1, stride=1, pad=0, dil=1, cus=0),      # 1x1, 64-column tiles
    dict(n=2, c=64, h=9, w=9, k=64, r=3, stride=2, pad=1, dil=1, cus=0),         # stride 2
]


def wgrad_ab(dev, n, c, h, w, k, r, stride, pad, dil, cus=0, seed=0):
    cp, kp = ops.rup4(c), ops.rup4(k)
    x = _nhwc(torch.randn(n, c, h, w, generator=_g(seed)), dev, cp)
    wt = torch.randn(k, c, r, r, generator=_g(seed + 1)).to(dev)
    d = ops.make_desc(x, wt, ops.ConvCfg(stride, pad, dil))
    dy = _nhwc(torch.randn(n, k, d.P, d.Q, generator=_g(seed + 6)), dev, kp)
    cnt = lambda name: int(_C.lib().up_conv_counter(name.encode()))
    try:
        _tune(cu_count=cus, glds32_wgrad=0)
        dw0, _ = ops.conv_bwd_weight_raw(x, dy, wt.shape, d, False)
        _tune(glds32_wgrad=1)
        w0, s0 = cnt("wgrad_glds32"), cnt("wgrad_glds32_st1")
        dw1, _ = ops.conv_bwd_weight_raw(x, dy, wt.shape, d, False)
        assert cnt("wgrad_glds32") == w0 + 1
        one_stage = cnt("wgrad_glds32_st1") > s0
    finally:
        _tune(**DEFAULTS)
    _same(dw1, dw0, "dw")
    return one_stage


BNRED = [
    dict(n=2, c=64, h=9, w=9, k=64, r=1, pad=0, dil=1, tile_want=100000),                 # 1x1, ragged last row tile
    dict(n=2, c=64, h=9, w=9, k=32, r=1, pad=0, dil=1, tile_want=100000, add=True),       # addend (identity-branch gradient) before the mask
    dict(n=2, c=64, h=9, w=9, k=32, r=1, pad=0, dil=1, tile_want=100000, add=True, mask_add=True),   # ... masked by the epilogue
    dict(n=3, c=128, h=7, w=7, k=64, r=1, pad=0, dil=1, tile_want=1, add=True, mask_add=True),       # 128x128 tile: operands fetched in the epilogue
    dict(n=3, c=128, h=7, w=7, k=64, r=3, pad=1, dil=1, tile_want=100000),                # 3x3, tap-sorted rows
    dict(n=3, c=128, h=7, w=7, k=64, r=3, pad=1, dil=1, tile_want=1),                     # 128x128 tile: operands fetched in the epilogue
    dict(n=3, c=128, h=7, w=7, k=64, r=3, pad=1, dil=1, tile_want=3),                     # 64x128 / 128x64
    dict(n=2, c=64, h=8, w=8, k=64, r=1, pad=0, dil=1, tile_want=100000, relu=False),     # BatchNorm without ReLU (down-sampling branch)
    dict(n=3, c=64, h=7, w=7, k=128, r=3, pad=1, dil=1, tile_want=100000, cus=4),          # with K-split tail tiles
]

# the real geometries of BASELINE configs[1] (368x368, B = 32) that carry the step (tile rule of the library: tile_want = 1500)
FULL = [
    dict(n=8, c=128, h=46, w=46, k=128, r=11, stride=1, pad=5, dil=1, tile_want=1500),                  # UniPose-LSTM head conv2 / conv3 at B = 8 (WIDE form)
    dict(n=32, c=256, h=23, w=23, k=256, r=3, stride=1, pad=1, dil=1, tile_want=1500, stats=True),      # layer3 conv2: 1060 tiles = 4 rounds + 36 K-split tails, tap-sorted
    dict(n=32, c=1024, h=23, w=23, k=256, r=1, stride=1, pad=0, dil=1, tile_want=1500, stats=True, add=True),   # layer3 conv1 (+ skip gradient in its data gradient)
    dict(n=32, c=256, h=23, w=23, k=1024, r=1, stride=1, pad=0, dil=1, tile_want=1500, stats=True),     # layer3 conv3: short reduction, 4240 tiles
    dict(n=32, c=256, h=23, w=23, k=256, r=3, stride=1, pad=18, dil=18, tile_want=1500, stats=True),    # WASP d = 18
    dict(n=32, c=512, h=23, w=23, k=512, r=3, stride=1, pad=4, dil=4, tile_want=1500, stats=True),      # layer4 d = 4
    dict(n=32, c=2048, h=23, w=23, k=512, r=1, stride=1, pad=0, dil=1, tile_want=1500, stats=True),     # layer4 conv1: 128-wide tiles
    dict(n=8, c=64, h=92, w=92, k=256, r=1, stride=1, pad=0, dil=1, tile_want=1500, stats=True),        # layer1 conv3 (B = 8)
    dict(n=8, c=128, h=46, w=46, k=128, r=3, stride=1, pad=1, dil=1, tile_want=1500, stats=True),       # layer2 conv2 (B = 8)
    dict(n=4, c=320, h=46, w=46, k=256, r=3, stride=1, pad=1, dil=1, tile_want=1500, affine=True, relu=True),   # decoder 3x3, folded epilogue
]

BNRED_FULL = [
    dict(n=32, c=256, h=23, w=23, k=1024, r=1, pad=0, dil=1, tile_want=1500),               # conv3's data gradient reduces bn2
    dict(n=32, c=256, h=23, w=23, k=256, r=3, pad=1, dil=1, tile_want=1500),                # conv2's (tap-sorted, K-split tails) reduces bn1
    dict(n=32, c=1024, h=23, w=23, k=256, r=1, pad=0, dil=1, tile_want=1500, add=True, mask_add=True),   # the next block's conv1 (+ masked skip gradient) reduces bn3
    dict(n=8, c=256, h=92, w=92, k=64, r=1, pad=0, dil=1, tile_want=1500, add=True),        # layer1: 128-wide tiles, operands fetched in the epilogue
    dict(n=32, c=512, h=23, w=23, k=512, r=3, pad=2, dil=2, tile_want=1500),                # layer4 d = 2
]
