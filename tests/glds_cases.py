"""A/B parity of the two bf16-storage implicit-GEMM generations (shared by the emulator and the GPU tests).

`igemm_glds_kernel` (unipose_amd/csrc/bf16s_glds.h: direct-to-LDS operand loads, 64-channel slices, tile-level tap
skipping, tap-sorted rows, LDS-transposed 16-byte stores) must reproduce the register-staged `igemm_bf16_kernel<HS>` of
rounds 1-2 EXACTLY: both accumulate the same bf16 products in the same k order on the same MFMA, skipped taps only
contribute exact zeros, the BatchNorm partials use the same arithmetic.  So outputs, data gradients and statistics are
compared for equality (the weight gradient of `wgrad_glds_kernel` against `wgrad_bf16_kernel<HS>` likewise: same pixel order
inside every split, fp32 slabs, same reduce pass) (== on floats: a skipped tap may turn a -0 into +0), not within a tolerance; the register-staged
kernel itself is pinned against torch in bf16s_cases.py."""
import ctypes as C

import torch

from unipose_amd import _C, ops

BF = torch.bfloat16


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
    return y.to(BF).to(dev)


def _same(a, b, what):
    a, b = a.float().cpu(), b.float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not torch.equal(a, b):
        d = (a - b).abs()
        raise AssertionError(f"{what}: {int((d > 0).sum())} of {d.numel()} elements differ, max |diff| {float(d.max()):.3e}, "
                             f"first at {tuple(int(i) for i in (d > 0).nonzero()[0])}")


def _merge(st):
    """(tiles, K, 3) Welford partials -> (count, mean, M2) per channel, merged in float64"""
    st = st.double().cpu()
    n, mean, m2 = st[0, :, 0].clone(), st[0, :, 1].clone(), st[0, :, 2].clone()
    for t in range(1, st.shape[0]):
        nb, mb, sb = st[t, :, 0], st[t, :, 1], st[t, :, 2]
        tot = n + nb
        d = mb - mean
        w = torch.where(tot > 0, nb / tot.clamp_min(1), torch.zeros_like(tot))
        mean = mean + d * w
        m2 = m2 + sb + d * d * n * w
        n = tot
    return n, mean, m2


def conv_ab(dev, n, c, h, w, k, r, stride, pad, dil, *, tile_want, stats=False, affine=False, residual=False, relu=False,
            add=False, seed=0, knob="glds", tune=None, expect_big=None):
    """forward (+ optional BatchNorm partials / folded epilogue / residual), data gradient (+ optional addend) and weight
    gradient of one convolution in bf16 storage: glds = 1 against glds = 0 under the same tile rule
    (knob="glds_big": the 8-wave (32 TM) x 256 tiles of bf16s_big.h against igemm_glds_kernel; `tune`: knobs held for the
    whole case; expect_big: launches that must have run on igemm_big_kernel with the knob on)."""
    cp, kp = ops.rup32(c), ops.rup32(k)
    x = _nhwc(torch.randn(n, c, h, w, generator=_g(seed)), dev, cp)
    wt = (torch.randn(k, c, r, r, generator=_g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5).to(dev)
    cfg = ops.ConvCfg(stride, pad, dil)
    kw = {}
    if affine:
        kw["scale"] = (0.5 + torch.rand(k, generator=_g(seed + 2))).to(dev)
        kw["shift"] = torch.randn(k, generator=_g(seed + 3)).to(dev)
        kw["bias"] = torch.randn(k, generator=_g(seed + 4)).to(dev)
    out = {}
    try:
        _tune(tile_want_bf16=tile_want, **(tune or {}))
        for mode in (1, 0):
            _tune(**{knob: mode})
            big0 = _C.lib().up_conv_counter(b"big")
            d0 = ops.make_desc(x, wt, cfg)
            res = None
            if residual:
                res = _nhwc(torch.randn(n, k, d0.P, d0.Q, generator=_g(seed + 5)), dev, kp)
            y, d, st = ops.conv_fwd_raw(x, wt, cfg, residual=res, relu=relu, stats=stats, **kw)
            dy = _nhwc(torch.randn(n, k, d.P, d.Q, generator=_g(seed + 6)), dev, kp)
            addt = _nhwc(torch.randn(n, c, h, w, generator=_g(seed + 7)), dev, cp) if add else None
            dx = ops.conv_bwd_data_raw(dy, wt, d, x.shape, x.device, add=addt) if stride == 1 else None
            dw, _ = ops.conv_bwd_weight_raw(x, dy, wt.shape, d, False)
            out[mode] = (y, st, dx, dw)
            if expect_big is not None:
                ran = _C.lib().up_conv_counter(b"big") - big0
                assert ran == (expect_big if mode else 0), ("launches on igemm_big_kernel", mode, ran)
    finally:
        _tune(glds=1, glds_big=1, tile_want_bf16=500, cu_count=0, big_min_k=1024)
    (y1, s1, dx1, dw1), (y0, s0, dx0, dw0) = out[1], out[0]
    _same(dw1, dw0, "dw")
    _same(y1, y0, "y")
    if stats:
        # per-tile partials (count, mean, M2): identical when both kernels tile the rows alike; with tap-sorted rows the
        # tiles hold different pixels, so the MERGED statistics are compared (float64 merge of the fp32 partials)
        m1, m0 = _merge(s1), _merge(s0)
        if torch.equal(s1.cpu()[..., 0], s0.cpu()[..., 0]) and r == 1:
            # same tiles: same accumulators, same formulas; the straight-line form for full tiles lets the compiler contract
            # multiply-adds differently on the GPU (measured: 0.4 % of the M2 values off by one ulp), the emulator agrees exactly
            err = float((s1.double().cpu() - s0.double().cpu()).abs().max() / s0.double().cpu().abs().max())
            assert err < 1e-6, ("BatchNorm partials", err)
        assert torch.equal(m1[0], m0[0]), "BatchNorm counts"
        for i, what in ((1, "mean"), (2, "M2")):
            err = float((m1[i] - m0[i]).abs().max() / m0[i].abs().max().clamp_min(1e-30))
            assert err < 1e-5, (what, err)
    if dx1 is not None:
        _same(dx1, dx0, "dx")
    return y1


# (n, c, h, w, k, r, stride, pad, dil, tile_want, flags)
SMALL = [
    dict(n=2, c=64, h=9, w=9, k=64, r=1, stride=1, pad=0, dil=1, tile_want=1, stats=True),          # 1x1, one slice, ragged row tile (162 rows)
    dict(n=3, c=64, h=7, w=7, k=128, r=3, stride=1, pad=1, dil=1, tile_want=1, stats=True),         # 128x128 tiles, 9 taps
    dict(n=3, c=64, h=7, w=7, k=128, r=3, stride=1, pad=1, dil=1, tile_want=100000, stats=True),    # 64x64 tiles
    dict(n=3, c=64, h=7, w=7, k=136, r=3, stride=1, pad=1, dil=1, tile_want=3),                     # 64x128 / 128x64 by the rule, ragged N (136)
    dict(n=4, c=64, h=7, w=7, k=64, r=3, stride=1, pad=3, dil=3, tile_want=1, stats=True),          # dilated: dead taps, tap-sorted rows
    dict(n=1, c=64, h=23, w=23, k=64, r=3, stride=1, pad=18, dil=18, tile_want=100000, stats=True),  # WASP d = 18 geometry
    dict(n=2, c=128, h=6, w=6, k=72, r=3, stride=1, pad=1, dil=1, tile_want=1, add=True),           # two slices per tap, dgrad addend, N = 72
    dict(n=2, c=64, h=8, w=8, k=64, r=1, stride=1, pad=0, dil=1, tile_want=1, affine=True, relu=True, residual=True),   # folded eval epilogue
    dict(n=2, c=64, h=8, w=8, k=64, r=3, stride=1, pad=2, dil=2, tile_want=1, affine=True, relu=True),   # eval, no residual, tap-sorted
    dict(n=2, c=64, h=9, w=9, k=64, r=3, stride=2, pad=1, dil=1, tile_want=1, stats=True),          # stride 2 forward
    dict(n=1, c=192, h=5, w=5, k=64, r=1, stride=1, pad=0, dil=1, tile_want=1),                     # three slices
]

# the real geometries of BASELINE configs[4] (736x736, B = 16) that carry the step
FULL = [
    dict(n=16, c=256, h=46, w=46, k=256, r=3, stride=1, pad=1, dil=1, tile_want=500, stats=True, add=True),      # layer3 conv2
    dict(n=16, c=1024, h=46, w=46, k=256, r=1, stride=1, pad=0, dil=1, tile_want=500, stats=True),               # layer3 conv1
    dict(n=16, c=256, h=46, w=46, k=1024, r=1, stride=1, pad=0, dil=1, tile_want=500, stats=True),               # layer3 conv3
    dict(n=16, c=256, h=46, w=46, k=256, r=3, stride=1, pad=18, dil=18, tile_want=500, stats=True),              # WASP d = 18
    dict(n=16, c=512, h=46, w=46, k=512, r=3, stride=1, pad=4, dil=4, tile_want=500, stats=True),                # layer4 d = 4
    dict(n=4, c=64, h=184, w=184, k=256, r=1, stride=1, pad=0, dil=1, tile_want=500, stats=True),                # layer1 conv3 (B = 4)
]


def bnred_case(dev, n, c, h, w, k, r, pad, dil, *, tile_want, add=False, mask_add=False, relu=True, seed=0, tune=None, expect_big=None):
    """bf16 storage: the data gradient of a convolution whose INPUT is z = relu(bn(y) (+ res)) also reduces that layer's
    BatchNorm-backward sums (up_conv2d_bwd_data_ex) — against up_bn_bwd's own reduce pass on the same dz;
    mask_add: the addend is an unmasked gradient whose ReLU mask the epilogue applies — against the pre-masked addend."""
    cp, kp = ops.rup32(c), ops.rup32(k)
    assert cp == c, "the fused forms need unpadded channel counts"
    L = _C.lib()
    wt = (torch.randn(k, c, r, r, generator=_g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5).to(dev)
    cfg = ops.ConvCfg(1, pad, dil)
    x = _nhwc(torch.randn(n, c, h, w, generator=_g(seed)), dev, cp)
    d = ops.make_desc(x, wt, cfg)
    dy = _nhwc(torch.randn(n, k, d.P, d.Q, generator=_g(seed + 6)), dev, kp)
    addt = _nhwc(torch.randn(n, c, h, w, generator=_g(seed + 7)), dev, cp) if add else None
    ybn = _nhwc(torch.randn(n, c, h, w, generator=_g(seed + 8)) * 2 + 0.5, dev, cp)
    rows = n * h * w

    def make_bits(s):
        pos = torch.rand(rows * c, generator=_g(s)) > 0.45
        words = torch.zeros((rows * c + 31) // 32 * 32, dtype=torch.int64)
        words[:rows * c] = pos.long()
        b = (words.view(-1, 32) << torch.arange(32)).sum(1)
        return torch.where(b >= 2 ** 31, b - 2 ** 32, b).to(torch.int32).to(dev), pos.view(n, h, w, c)

    bits, _ = make_bits(seed + 9) if relu else (None, None)
    abits, apos = make_bits(seed + 13) if (add and mask_add) else (None, None)
    mean = torch.randn(c, generator=_g(seed + 10)).to(dev)
    invstd = (0.5 + torch.rand(c, generator=_g(seed + 11))).to(dev)
    gamma = (0.5 + torch.rand(c, generator=_g(seed + 12))).to(dev)
    try:
        _tune(tile_want_bf16=tile_want, glds=1, **(tune or {}))
        big0 = _C.lib().up_conv_counter(b"big")
        slot = ops.BnSlot(ybn, bits, mean, invstd, c)
        dx = ops.conv_bwd_data_raw(dy, wt, d, x.shape, x.device, add=addt, bn_slot=slot, add_bits=abits)
        assert slot.partial is not None, "the launch did not take the fused reduction"
        pre = addt if abits is None else (addt.float() * apos.to(dev).float()).to(BF)
        dx0 = ops.conv_bwd_data_raw(dy, wt, d, x.shape, x.device, add=pre)
        if expect_big is not None:
            assert _C.lib().up_conv_counter(b"big") - big0 == expect_big, "launches on igemm_big_kernel"
    finally:
        _tune(glds=1, glds_big=1, tile_want_bf16=500, cu_count=0, big_min_k=1024)
    _same(dx, dx0, "dx (fused reduction / masked addend on / off)")
    outs = []
    for fused in (True, False):
        dyb, dres = torch.empty_like(ybn), torch.empty_like(ybn)
        dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
        ws = ops.workspace(dx.device, L.up_bn_bwd_workspace(rows, c))
        pb = None if bits is None else bits.data_ptr()
        if fused:
            _C.check(L.up_bn_bwd_prereduced_t(dx.data_ptr(), cp, pb, ybn.data_ptr(), cp, gamma.data_ptr(), mean.data_ptr(),
                                              invstd.data_ptr(), int(relu), 1, dyb.data_ptr(), cp, dres.data_ptr(), cp,
                                              dgb[0].data_ptr(), dgb[1].data_ptr(), None, None, slot.partial.data_ptr(),
                                              slot.partial.shape[0], rows, c, 1, ops._stream(dx)), "bn_bwd_prereduced")
        else:
            _C.check(L.up_bn_bwd_acc_t(dx.data_ptr(), cp, None, 0, pb, ybn.data_ptr(), cp, gamma.data_ptr(), mean.data_ptr(),
                                       invstd.data_ptr(), int(relu), 1, dyb.data_ptr(), cp, dres.data_ptr(), cp, dgb[0].data_ptr(),
                                       dgb[1].data_ptr(), None, None, ws.data_ptr(), ws.numel(), rows, c, 1, ops._stream(dx)), "bn_bwd")
        outs.append((dyb.float().cpu(), dres.float().cpu(), dgb.cpu()))
    (dy1, dr1, g1), (dy0, dr0, g0) = outs
    _same(dr1, dr0, "dres")
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 2e-5 * max(scale, 1.0), ("dgamma / dbeta", float((g1 - g0).abs().max()), scale)
    # dy is rounded to bf16: one ulp where the re-ordered sums moved a value across a rounding boundary
    assert float((dy1 - dy0).abs().max()) <= 2.0 ** -7 * float(dy0.abs().max()), ("dy", float((dy1 - dy0).abs().max()))
    assert float(((dy1 - dy0).abs() > 0).float().mean()) < 0.02
    return dx


BNRED = [
    dict(n=2, c=64, h=9, w=9, k=64, r=1, pad=0, dil=1, tile_want=100000),                              # 1x1, ragged last row tile
    dict(n=2, c=64, h=9, w=9, k=32, r=1, pad=0, dil=1, tile_want=100000, add=True),                    # addend: fp32 half-image path
    dict(n=2, c=64, h=9, w=9, k=32, r=1, pad=0, dil=1, tile_want=100000, add=True, mask_add=True),     # addend masked in the epilogue
    dict(n=3, c=128, h=7, w=7, k=64, r=3, pad=1, dil=1, tile_want=1),                                  # 3x3, tap-sorted rows, 128x128 tile
    dict(n=3, c=128, h=7, w=7, k=64, r=3, pad=1, dil=1, tile_want=3),                                  # 64x128 / 128x64
    dict(n=2, c=64, h=8, w=8, k=64, r=1, pad=0, dil=1, tile_want=100000, relu=False),                  # BatchNorm without ReLU
]
BNRED_FULL = [
    dict(n=16, c=256, h=46, w=46, k=1024, r=1, pad=0, dil=1, tile_want=500),                           # conv3's data gradient reduces bn2
    dict(n=16, c=256, h=46, w=46, k=256, r=3, pad=1, dil=1, tile_want=500),                            # conv2's (tap-sorted) reduces bn1
    dict(n=16, c=1024, h=46, w=46, k=256, r=1, pad=0, dil=1, tile_want=500, add=True, mask_add=True),  # next block's conv1 reduces bn3
]


# igemm_big_kernel (bf16s_big.h) against igemm_glds_kernel: knob glds_big; cu_count = 1 makes the tile rule pick the row count
# that needs the fewest rows in total (M = 160 / 192 / 256 -> TM = 5 / 6 / 8), big_min_k lets short reductions through
_BIG = dict(knob="glds_big", tile_want=1)
BIG_SMALL = [
    dict(n=2, c=64, h=9, w=9, k=256, r=1, stride=1, pad=0, dil=1, stats=True, tune=dict(big_min_k=64), expect_big=1, **_BIG),            # 162 rows: a full 160-row tile + 2 rows
    dict(n=3, c=64, h=8, w=8, k=256, r=1, stride=1, pad=0, dil=1, stats=True, tune=dict(big_min_k=64, cu_count=1), expect_big=1, **_BIG),  # 192 rows: TM = 6
    dict(n=4, c=128, h=8, w=8, k=256, r=1, stride=1, pad=0, dil=1, stats=True, add=True, tune=dict(big_min_k=64, cu_count=1), expect_big=1, **_BIG),  # 256 rows: TM = 8, two slices, addend (dgrad N = 128: old kernel)
    dict(n=3, c=64, h=7, w=7, k=256, r=3, stride=1, pad=1, dil=1, stats=True, expect_big=1, **_BIG),                                     # 3x3, tap-sorted rows, ragged tile (147 rows)
    dict(n=4, c=128, h=7, w=7, k=512, r=3, stride=1, pad=3, dil=3, stats=True, expect_big=1, **_BIG),                                    # dilated: dead taps, two column tiles
    dict(n=2, c=64, h=9, w=9, k=256, r=1, stride=1, pad=0, dil=1, affine=True, relu=True, residual=True, tune=dict(big_min_k=64), expect_big=1, **_BIG),   # folded eval epilogue + residual
    dict(n=2, c=64, h=9, w=9, k=256, r=3, stride=1, pad=2, dil=2, affine=True, relu=True, expect_big=1, **_BIG),                         # eval, no residual, tap-sorted
    dict(n=2, c=64, h=17, w=17, k=256, r=3, stride=2, pad=1, dil=1, stats=True, expect_big=1, **_BIG),                                   # stride-2 forward
    dict(n=2, c=256, h=9, w=9, k=256, r=1, stride=1, pad=0, dil=1, stats=True, add=True, expect_big=2, **_BIG),                          # data gradient on the big tiles too, with addend
]
BIG_BNRED = [
    dict(n=2, c=256, h=9, w=9, k=64, r=1, pad=0, dil=1, tile_want=1, tune=dict(big_min_k=64), expect_big=2),                               # 1x1, ragged last tile
    dict(n=2, c=256, h=9, w=9, k=64, r=1, pad=0, dil=1, tile_want=1, add=True, tune=dict(big_min_k=64), expect_big=2),                     # addend: fp32 image path
    dict(n=2, c=256, h=9, w=9, k=64, r=1, pad=0, dil=1, tile_want=1, add=True, mask_add=True, tune=dict(big_min_k=64), expect_big=2),      # addend masked in the epilogue
    dict(n=3, c=256, h=7, w=7, k=64, r=3, pad=1, dil=1, tile_want=1, expect_big=2),                                                        # 3x3, tap-sorted rows
    dict(n=3, c=256, h=8, w=8, k=128, r=1, pad=0, dil=1, tile_want=1, relu=False, tune=dict(big_min_k=64, cu_count=1), expect_big=2),      # TM = 6, BatchNorm without ReLU
]

BIG_FULL = [
    dict(n=16, c=256, h=46, w=46, k=256, r=3, stride=1, pad=1, dil=1, stats=True, add=True, expect_big=2, **_BIG),      # layer3 conv2: 212 tiles of 160 rows
    dict(n=16, c=1024, h=46, w=46, k=256, r=1, stride=1, pad=0, dil=1, stats=True, expect_big=2, **_BIG),               # layer3 conv1; its data gradient: N = 1024, K = 256
    dict(n=16, c=256, h=46, w=46, k=1024, r=1, stride=1, pad=0, dil=1, stats=True, expect_big=2, **_BIG),               # layer3 conv3: 708 tiles of 192 rows
    dict(n=16, c=256, h=46, w=46, k=256, r=3, stride=1, pad=18, dil=18, stats=True, expect_big=2, **_BIG),              # WASP d = 18
    dict(n=16, c=512, h=46, w=46, k=512, r=3, stride=1, pad=4, dil=4, stats=True, expect_big=2, **_BIG),                # layer4 d = 4
    dict(n=5, c=256, h=46, w=46, k=256, r=3, stride=1, pad=1, dil=1, stats=True, expect_big=2, **_BIG),                 # fewer tiles than CUs, ragged last tile
]
BIG_BNRED_FULL = [dict(c, expect_big=2) for c in BNRED_FULL]
# (the library's default sends reductions shorter than 1024 to igemm_glds_kernel: these cases exercise igemm_big_kernel at every length)
for _c in BIG_SMALL + BIG_FULL + BIG_BNRED + BIG_BNRED_FULL:
    _c["tune"] = dict(dict(big_min_k=64), **(_c.get("tune") or {}))
