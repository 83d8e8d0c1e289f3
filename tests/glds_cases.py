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
            add=False, seed=0, kt=64, st=2, wkp=64, wst=2, split=0, cus=0, big=0):
    """forward (+ optional BatchNorm partials / folded epilogue / residual) and data gradient (+ optional addend) of one
    convolution in bf16 storage: glds = 1 against glds = 0 under the same tile rule.
    split = 1: the K-split of tail tiles is on in the glds kernels (`cus` shrinks the chip so that a small launch has whole rounds of
    tiles + a tail); a split tile adds its shares in a different order, so outputs are then compared within fp32 round-off of the
    reduction instead of exactly.
    big = n: launches that would run at least n 256 x 128 tiles use them (glds_256) instead of 128 x 128 ones."""
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
        _tune(tile_want_bf16=tile_want, glds_kt=kt, glds_st=st, wgrad_kp=wkp, wgrad_st=wst, glds_split=split, cu_count=cus, glds_split_q=4, glds_split_maxp=256, glds_256=big)
        for mode in (1, 0):
            _tune(glds=mode)
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
    finally:
        _tune(glds=1, tile_want_bf16=500, glds_kt=32, glds_st=2, wgrad_kp=64, wgrad_st=2, glds_split=0, cu_count=0, glds_split_q=2, glds_split_maxp=4, glds_256=0)
    (y1, s1, dx1, dw1), (y0, s0, dx0, dw0) = out[1], out[0]
    _same(dw1, dw0, "dw")
    if split:
        # bf16 outputs of fp32 sums that differ by re-association: at most one bf16 ulp (2^-8 relative) on a few elements
        for a_, b_, what in ((y1, y0, "y"), (dx1, dx0, "dx")):
            if a_ is None:
                continue
            a_, b_ = a_.float().cpu(), b_.float().cpu()
            err = (a_ - b_).abs()
            assert float((err / b_.abs().clamp_min(1e-2)).max()) <= 2.0 ** -7, (what, float(err.max()))
            assert float((err > 0).float().mean()) < 0.02, (what, "too many elements differ", float((err > 0).float().mean()))
        if stats:
            m1, m0 = _merge(s1), _merge(s0)
            assert torch.equal(m1[0], m0[0]), "BatchNorm counts"
            for i, what in ((1, "mean"), (2, "M2")):
                err = float((m1[i] - m0[i]).abs().max() / m0[i].abs().max().clamp_min(1e-30))
                assert err < 1e-4, (what, err)
        return y1
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

# K-split tail tiles (chip shrunk to `cus` CUs): whole rounds + tail parts in one launch
SPLIT = [
    dict(n=3, c=64, h=7, w=7, k=128, r=3, stride=1, pad=1, dil=1, tile_want=100000, stats=True, cus=4),   # 64x64 tiles: 3 x 2 = 6 tiles = 4 + 2 tails x 2 parts
    dict(n=2, c=128, h=6, w=6, k=72, r=3, stride=1, pad=1, dil=1, tile_want=1, add=True, cus=0),          # one 128x128 tile, 36 slices -> 18 parts, addend after the merge
    dict(n=4, c=64, h=7, w=7, k=64, r=3, stride=1, pad=3, dil=3, tile_want=100000, stats=True, cus=3),    # tap-sorted, tiles with different live taps: 4 tiles = 3 + 1 x 3
    dict(n=2, c=64, h=9, w=9, k=64, r=1, stride=1, pad=0, dil=1, tile_want=100000, stats=True, cus=2),    # 1x1, two slices: 3 tiles = 2 + 1 tail; p = 1 (too short): no split
    dict(n=1, c=256, h=5, w=5, k=64, r=1, stride=1, pad=0, dil=1, tile_want=1, affine=True, relu=True, residual=True, cus=0),   # 1x1 8 slices -> 4 parts, folded epilogue
    dict(n=1, c=64, h=5, w=5, k=64, r=3, stride=1, pad=6, dil=6, tile_want=1, stats=True, cus=0),         # only the centre tap lives: 2 live slices under 9 parts -> empty shares
]

# 256 x 128 tiles (kt = 32, two stages): ragged last tiles whose second half is partly or wholly past the end, BatchNorm partial rows per
# 128-row half, tap-sorted rows, folded epilogue with residual (fp32 half-image path), data-gradient addend
BIG = [
    dict(n=3, c=64, h=7, w=7, k=128, r=3, stride=1, pad=1, dil=1, tile_want=1, stats=True),                 # 147 rows: one tile, second half 19 rows
    dict(n=4, c=64, h=14, w=14, k=136, r=1, stride=1, pad=0, dil=1, tile_want=1, stats=True),               # 784 rows: 4 tiles, the last with 16 rows; ragged N
    dict(n=5, c=64, h=10, w=10, k=128, r=3, stride=1, pad=2, dil=2, tile_want=1, stats=True, add=True),      # 500 rows, tap-sorted, dgrad addend
    dict(n=2, c=128, h=16, w=16, k=128, r=1, stride=1, pad=0, dil=1, tile_want=1, affine=True, relu=True, residual=True),   # 512 rows = 2 full tiles
    dict(n=4, c=64, h=8, w=8, k=128, r=3, stride=1, pad=3, dil=3, tile_want=1, affine=True, relu=True),      # 256 rows exactly, eval epilogue
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
