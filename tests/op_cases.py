"""Per-operator parity cases shared by the CPU-emulation tests and the GPU tests.  The checker is plain
torch fp32 on CPU (the ops the reference itself calls: F.conv2d, F.batch_norm, ...)."""
import torch
import torch.nn.functional as F

from unipose_amd import ops


def nhwc(x, dev, pad_to=None):
    """NCHW cpu tensor -> NHWC (channel-padded to a multiple of 4) on dev."""
    n, c, h, w = x.shape
    cp = pad_to or ops.rup4(c)
    y = torch.zeros(n, h, w, cp)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.to(dev)


def nchw(y, c):
    return y.detach().cpu()[..., :c].permute(0, 3, 1, 2).contiguous()


def rel(a, b):
    a, b = a.double(), b.double()
    s = b.abs().max().item()
    return (a - b).abs().max().item() / (s if s > 0 else 1.0)


def g(seed):
    gen = torch.Generator()
    gen.manual_seed(seed)
    return gen


def conv_case(dev, n, c, h, w, k, r, stride, pad, dil, bias=False, relu=False, seed=0, tol=2e-5):
    x = torch.randn(n, c, h, w, generator=g(seed))
    wt = torch.randn(k, c, r, r, generator=g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5
    b = torch.randn(k, generator=g(seed + 2)) if bias else None
    xd = nhwc(x, dev).requires_grad_(True)
    wd = wt.clone().to(dev).requires_grad_(True)
    bd = b.clone().to(dev).requires_grad_(True) if bias else None
    y = ops.ConvBias.apply(xd, wd, bd, ops.ConvCfg(stride, pad, dil), relu)

    xr, wr = x.detach().clone().requires_grad_(True), wt.detach().clone().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
    if relu:
        # ReLU is discontinuous: an element within fp32 round-off of 0 may land on either side.  The
        # forward values are compared as usual; for the BACKWARD comparison the reference graph uses the
        # device's own mask so that one flipped element does not masquerade as a gradient error.
        yr_fwd = F.relu(yr)
        yr = yr * (nchw(y, k) > 0).float()
    else:
        yr_fwd = yr
    dy = torch.randn(yr.shape, generator=g(seed + 3))
    yr.backward(dy)
    assert y.shape == (n, yr.shape[2], yr.shape[3], ops.rup4(k))
    y.backward(nhwc(dy, dev))
    yr = yr_fwd
    errs = {
        "y": rel(nchw(y, k), yr.detach()),
        "dx": rel(nchw(xd.grad, c), xr.grad),
        "dw": rel(wd.grad.cpu(), wr.grad),
    }
    if bias:
        errs["db"] = rel(bd.grad.cpu(), br.grad)
    if ops.rup4(k) != k:
        assert float(y.detach()[..., k:].abs().max()) == 0.0
    bad = {k_: v for k_, v in errs.items() if not v < tol}
    assert not bad, (bad, errs)
    return errs


def dgrad_add_case(dev, n, c, h, w, k, r, stride, pad, dil, seed=0, tol=2e-5):
    """up_conv2d_bwd_data with its optional addend: dx = conv_transpose(dy) + add (the skip-connection gradient)."""
    wt = torch.randn(k, c, r, r, generator=g(seed)) * (2.0 / (c * r * r)) ** 0.5
    x = torch.zeros(n, c, h, w, requires_grad=True)
    yr = F.conv2d(x, wt, stride=stride, padding=pad, dilation=dil)
    dy = torch.randn(yr.shape, generator=g(seed + 1))
    add = torch.randn(n, c, h, w, generator=g(seed + 2))
    yr.backward(dy)
    xd = nhwc(x.detach(), dev)
    d = ops.make_desc(xd, wt.to(dev), ops.ConvCfg(stride, pad, dil))
    dx = ops.conv_bwd_data_raw(nhwc(dy, dev), wt.to(dev), d, xd.shape, xd.device, add=nhwc(add, dev))
    e = rel(nchw(dx, c), x.grad + add)
    assert e < tol, e
    if ops.rup4(c) != c:
        assert float(dx[..., c:].abs().max()) == 0.0
    return e


def conv_bn_case(dev, n, c, h, w, k, r, stride, pad, dil, relu=True, residual=False, train=True, seed=0, tol=5e-5):
    x = torch.randn(n, c, h, w, generator=g(seed)) + 0.3
    conv = torch.nn.Conv2d(c, k, r, stride=stride, padding=pad, dilation=dil, bias=False)
    bn = torch.nn.BatchNorm2d(k)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5)
        bn.weight.copy_(0.5 + torch.rand(k, generator=g(seed + 2)))
        bn.bias.copy_(0.2 * torch.randn(k, generator=g(seed + 3)))
        bn.running_mean.copy_(0.1 * torch.randn(k, generator=g(seed + 4)))
        bn.running_var.copy_(0.5 + torch.rand(k, generator=g(seed + 5)))
    import copy
    conv_d, bn_d = copy.deepcopy(conv).to(dev), copy.deepcopy(bn).to(dev)
    conv.train(train), bn.train(train), conv_d.train(train), bn_d.train(train)
    xr = x.clone().requires_grad_(True)
    yr = bn(conv(xr))
    res = torch.randn(yr.shape, generator=g(seed + 6)) if residual else None
    rr = res.clone().requires_grad_(True) if residual else None
    if residual:
        yr = yr + rr
    xd = nhwc(x, dev).requires_grad_(True)
    rd = nhwc(res, dev).requires_grad_(True) if residual else None
    y = ops.conv_bn_act(xd, conv_d, bn_d, relu=relu, residual=rd)
    yr_fwd = F.relu(yr) if relu else yr
    if relu:
        yr = yr * (nchw(y, k) > 0).float()      # device mask for the backward comparison (see conv_case)
    dy = torch.randn(yr.shape, generator=g(seed + 7))
    yr.backward(dy)
    y.backward(nhwc(dy, dev))
    yr = yr_fwd
    errs = {
        "y": rel(nchw(y, k), yr.detach()),
        "dx": rel(nchw(xd.grad, c), xr.grad),
        "dw": rel(conv_d.weight.grad.cpu(), conv.weight.grad),
        "dgamma": rel(bn_d.weight.grad.cpu(), bn.weight.grad),
        "dbeta": rel(bn_d.bias.grad.cpu(), bn.bias.grad),
        "rm": rel(bn_d.running_mean.cpu(), bn.running_mean),
        "rv": rel(bn_d.running_var.cpu(), bn.running_var),
    }
    if residual:
        errs["dres"] = rel(nchw(rd.grad, k), rr.grad)
    assert int(bn_d.num_batches_tracked) == int(bn.num_batches_tracked)
    bad = {k_: v for k_, v in errs.items() if not v < tol}
    assert not bad, (bad, errs)
    # inference fast path (BN folded into the conv epilogue) against eval-mode torch
    conv.eval(), bn.eval(), conv_d.eval(), bn_d.eval()
    with torch.no_grad():
        ye = bn(conv(x))
        if residual:
            ye = ye + res
        if relu:
            ye = F.relu(ye)
        yf = ops.conv_bn_act(nhwc(x, dev), conv_d, bn_d, relu=relu, residual=nhwc(res, dev) if residual else None)
    e = rel(nchw(yf, k), ye)
    assert e < tol, ("eval fused", e)
    return errs


def layout_case(dev):
    x = torch.randn(2, 3, 5, 7, generator=g(1))
    xd = x.to(dev).requires_grad_(True)
    y = ops.ToNHWC.apply(xd)
    assert y.shape == (2, 5, 7, 4)
    assert rel(nchw(y, 3), x) == 0 and float(y.detach()[..., 3].abs().max()) == 0
    z = ops.ToNCHW.apply(y, 3)
    assert torch.equal(z.detach().cpu(), x)
    dz = torch.randn(z.shape, generator=g(2))
    z.backward(dz.to(dev))
    assert torch.equal(xd.grad.cpu(), dz)


def maxpool_case(dev, n=2, c=8, h=9, w=10):
    x = torch.randn(n, c, h, w, generator=g(3))
    x[0, 0, :3, :3] = 1.5          # ties: first max in window-scan order must win
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(yr.shape, generator=g(4))
    yr.backward(dy)
    xd = nhwc(x, dev).requires_grad_(True)
    y = ops.MaxPool3s2.apply(xd)
    y.backward(nhwc(dy, dev))
    assert torch.equal(nchw(y, c), yr.detach())
    assert rel(nchw(xd.grad, c), xr.grad) < 1e-6


def bilinear_case(dev, n, c, h, w, p, q, tol=1e-5):
    x = torch.randn(n, c, h, w, generator=g(5))
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(p, q), mode="bilinear", align_corners=True)
    dy = torch.randn(yr.shape, generator=g(6))
    yr.backward(dy)
    xd = nhwc(x, dev).requires_grad_(True)
    y = ops.Bilinear.apply(xd, p, q)
    y.backward(nhwc(dy, dev))
    assert rel(nchw(y, c), yr.detach()) < tol
    assert rel(nchw(xd.grad, c), xr.grad) < tol


def gap_case(dev, n=3, c=72, h=5, w=7):
    x = torch.randn(n, c, h, w, generator=g(7))
    xr = x.clone().requires_grad_(True)
    yr = F.adaptive_avg_pool2d(xr, 1)
    dy = torch.randn(yr.shape, generator=g(8))
    yr.backward(dy)
    xd = nhwc(x, dev).requires_grad_(True)
    y = ops.GlobalAvgPool.apply(xd)
    y.backward(nhwc(dy, dev))
    assert rel(nchw(y, c), yr.detach()) < 1e-5
    assert rel(nchw(xd.grad, c), xr.grad) < 1e-6


def concat_case(dev):
    a = torch.randn(2, 8, 4, 5, generator=g(9))
    b = torch.randn(2, 12, 4, 5, generator=g(10))
    ad, bd = nhwc(a, dev).requires_grad_(True), nhwc(b, dev).requires_grad_(True)
    y = ops.ConcatC.apply(0, ad, bd)
    assert torch.equal(nchw(y, 20), torch.cat((a, b), 1))
    dy = torch.randn(2, 20, 4, 5, generator=g(11))
    y.backward(nhwc(dy, dev))
    assert torch.equal(nchw(ad.grad, 8), dy[:, :8]) and torch.equal(nchw(bd.grad, 12), dy[:, 8:])


def dropout_case(dev):
    x = torch.randn(2, 6, 6, 16, generator=g(12)).to(dev).requires_grad_(True)
    m = (torch.rand(2, 6, 6, 16, generator=g(13)) > 0.5).float().to(dev)
    y = ops.Dropout.apply(x, 0.5, 1, m)
    assert torch.equal(y.detach().cpu(), (x.detach() * m / 0.5).cpu())
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad.cpu(), (m / 0.5).cpu())
    big = torch.ones(64, 8, 8, 64).to(dev)
    y1 = ops.Dropout.apply(big, 0.3, 7, None)
    y2 = ops.Dropout.apply(big, 0.3, 7, None)
    y3 = ops.Dropout.apply(big, 0.3, 8, None)
    keep = float((y1 != 0).float().mean())
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    assert abs(keep - 0.7) < 0.01, keep
    assert abs(float(y1.max()) - 1 / 0.7) < 1e-6


def mse_case(dev):
    y = torch.randn(2, 17, 9, 9, generator=g(14))
    t = torch.rand(2, 17, 9, 9, generator=g(15))
    yr = y.clone().requires_grad_(True)
    lr = F.mse_loss(yr, t)
    (lr * 3.0).backward()
    yd = y.to(dev).requires_grad_(True)
    l = ops.mse_loss(yd, t.to(dev))
    (l * 3.0).backward()
    assert abs(float(l) - float(lr)) < 1e-6 * abs(float(lr))
    assert rel(yd.grad.cpu(), yr.grad) < 1e-6


def avgpool_case(dev, h=40, w=48):
    c = torch.rand(2, 1, h, w, generator=g(16))
    ref = F.avg_pool2d(c, 9, 8, 1)
    p, q = ref.shape[2:]
    buf = torch.zeros(2, p, q, 16).to(dev)
    ops.avgpool9s8_into(c.to(dev), buf, 14)
    assert rel(buf.cpu()[..., 14], ref[:, 0]) < 1e-6
    assert float(buf[..., :14].abs().max()) == 0 and float(buf[..., 15].abs().max()) == 0


def lstm_case(dev):
    cg, n, h, w = 15, 2, 5, 6
    G0 = torch.randn(n, h, w, 48, generator=g(17))
    G = torch.randn(n, h, w, 60, generator=g(18))
    cp = torch.randn(n, h, w, 16, generator=g(19))
    dcell = torch.randn(n, h, w, 16, generator=g(20))
    dhide = torch.randn(n, h, w, 16, generator=g(21))
    # reference math (model/uniposeLSTM.py:17-22, 41-62) with torch autograd
    a = G0.clone().requires_grad_(True)
    gg, ii, oo = torch.tanh(a[..., :cg]), torch.sigmoid(a[..., cg:2 * cg]), torch.sigmoid(a[..., 2 * cg:3 * cg])
    cell = torch.tanh(gg * ii)
    hide = oo * cell
    (cell * dcell[..., :cg]).sum().backward(retain_graph=True)
    (hide * dhide[..., :cg]).sum().backward()
    ad = G0.to(dev).requires_grad_(True)
    c_, h_ = ops.LSTM0Gates.apply(ad, cg)
    ((c_ * dcell.to(dev)).sum() + (h_ * dhide.to(dev)).sum()).backward()
    assert rel(c_.detach().cpu()[..., :cg], cell.detach()) < 1e-5 and rel(h_.detach().cpu()[..., :cg], hide.detach()) < 1e-5
    assert rel(ad.grad.cpu()[..., :45], a.grad[..., :45]) < 1e-5
    b = G.clone().requires_grad_(True)
    cpr = cp.clone().requires_grad_(True)
    gg, ii = torch.tanh(b[..., :cg]), torch.sigmoid(b[..., cg:2 * cg])
    oo, ff = torch.sigmoid(b[..., 2 * cg:3 * cg]), torch.sigmoid(b[..., 3 * cg:4 * cg])
    cell = ff * cpr[..., :cg] + ii * gg
    hide = oo * torch.tanh(cell)
    ((cell * dcell[..., :cg]).sum() + (hide * dhide[..., :cg]).sum()).backward()
    bd = G.to(dev).requires_grad_(True)
    cpd = cp.to(dev).requires_grad_(True)
    c_, h_ = ops.LSTMGates.apply(bd, cpd, cg)
    ((c_ * dcell.to(dev)).sum() + (h_ * dhide.to(dev)).sum()).backward()
    assert rel(c_.detach().cpu()[..., :cg], cell.detach()) < 1e-5 and rel(h_.detach().cpu()[..., :cg], hide.detach()) < 1e-5
    assert rel(bd.grad.cpu(), b.grad) < 1e-5
    assert rel(cpd.grad.cpu()[..., :cg], cpr.grad[..., :cg]) < 1e-5


def argmax_case(dev, golden_dir):
    import os
    import numpy as np
    gd = np.load(os.path.join(golden_dir, "g6_argmax.npz"))
    preds, mx, idx = ops.heatmap_argmax(torch.from_numpy(gd["hm"]).to(dev))
    assert np.array_equal(preds.cpu().numpy(), gd["preds"])          # bit-exact vs the reference's numpy
    assert np.array_equal(mx.cpu().numpy(), gd["maxvals"])
    flat = gd["hm"].reshape(3, 15, -1)
    assert np.array_equal(idx.cpu().numpy(), flat.argmax(2).astype(np.int32))
    hm = torch.from_numpy(gd["hm"][:1]).to(dev)
    kp = ops.get_kpts(hm)
    ref = []
    for m in gd["hm"][0][1:]:
        r, c = np.unravel_index(m.argmax(), m.shape)
        ref.append([int(c * 368.0 / 46), int(r * 368.0 / 46)])
    assert kp == ref


def accuracy_case(dev, golden_dir):
    """ops.accuracy (device argmax + PCK kernel) against the reference's own results (G7)."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "g7_accuracy.npz"))
    for ds in ("LSP", "MPII", "Penn_Action"):
        out = torch.from_numpy(g[ds + "_out"]).to(dev)
        tgt = torch.from_numpy(g[ds + "_tgt"]).to(dev)
        for tag in ("std", "tight"):
            tk, th = (float(v) for v in g[f"{ds}_{tag}_thr"])
            acc, pck, pckh, cnt, pred, vis = ops.accuracy(out, tgt, tk, th, ds)
            k = f"{ds}_{tag}_"
            assert np.array_equal(pred, g[k + "pred"]) and cnt == int(g[k + "cnt"]), (ds, tag)
            assert np.array_equal(vis, g[k + "vis"]), (ds, tag)
            for name, got in (("acc", acc), ("pck", pck), ("pckh", pckh)):
                assert np.array_equal(got, g[k + name]), (ds, tag, name, got, g[k + name])


def targets_case(dev, golden_dir):
    """ops.make_heatmaps / make_centermaps against maps built with the reference's own Gaussian (G8): bit-exact float32
    wherever the device exp() agrees with numpy's to the last float64 bit after rounding to float32, which the
    comparison allows one float32 ulp for; the 0.0099 cut and the clip are compared exactly away from the cut."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "g8_targets.npz"))

    def check(got, ref, what):
        got = got.cpu().numpy()
        assert got.shape == ref.shape and got.dtype == np.float32, what
        cut = np.float32(0.0099)
        # a device exp() one float64 ulp from numpy's may fall on the other side of the 0.0099 cut: tolerate exactly that
        flipped = ((got == 0) & (np.abs(ref - cut) < 1e-6)) | ((ref == 0) & (np.abs(got - cut) < 1e-6))
        diff = np.abs(got - ref)[~flipped]
        assert diff.size and float(diff.max()) <= 1.2e-7, (what, float(diff.max()))       # one float32 ulp below 1
        assert int(flipped.sum()) <= 2, (what, int(flipped.sum()))

    for tag in ("lsp", "penn_sigma1", "odd_stride4"):
        stride, sigma = (float(v) for v in g[tag + "_cfg"])
        hm = ops.make_heatmaps(g[tag + "_kpt"], 368, 368, stride, sigma, dev)
        check(hm, g[tag + "_hm"], tag)
    cm = ops.make_centermaps(g["centers"], 96, 80, 3.0, dev)
    check(cm, g["centermaps"], "centermaps")


def normalize_case(dev):
    img = torch.randint(0, 256, (3, 37, 41, 3), generator=torch.Generator().manual_seed(5)).float()
    got = ops.normalize_image(img.to(dev))
    ref = img.permute(0, 3, 1, 2).contiguous().sub(128.0).div(256.0)      # Mytransforms.to_tensor + normalize
    assert torch.equal(got.cpu(), ref)


def bn_large_mean_case(dev, rows=200, c=64, seed=21):
    """The normalisation pass with |mean| >> std (the reference's wasp.global_avg_pool BatchNorm: |mean| / std = 364 on the G14 input).
    up_bn_apply_centered_t evaluates (y - mean) * scale + beta like ATen: every element within an ulp or two of the float64 value
    of that expression over the SAME fp32 coefficient vectors, while y * scale + shift (up_bn_apply_t, kept for folded / legacy
    callers) is off by 2^-24 |mean| scale per element — random errors that flip ReLU decisions the reference does not flip."""
    import ctypes as C
    from unipose_amd import _C
    gen = g(seed)
    mean = (3000 + 3000 * torch.rand(c, generator=gen)).float()
    std = 0.5 + torch.rand(c, generator=gen)
    y = (mean.view(1, -1) + std.view(1, -1) * torch.randn(rows, c, generator=gen)).float()
    scale = ((0.5 + torch.rand(c, generator=gen)) / std).float()
    beta = (0.2 * torch.randn(c, generator=gen)).float()
    shift = (beta.double() - mean.double() * scale.double()).float()
    z64 = (y.double() - mean.double().view(1, -1)) * scale.double().view(1, -1) + beta.double().view(1, -1)
    yd, md, sd_, bd, hd = (t.to(dev).contiguous() for t in (y, mean, scale, beta, shift))
    zc, zf = torch.empty_like(yd), torch.empty_like(yd)
    L = _C.lib()
    st = ops._stream(yd)
    _C.check(L.up_bn_apply_centered_t(yd.data_ptr(), c, md.data_ptr(), sd_.data_ptr(), bd.data_ptr(), None, 0, 0, zc.data_ptr(), c,
                                      None, rows, c, 0, st), "bn_apply_centered")
    _C.check(L.up_bn_apply_t(yd.data_ptr(), c, sd_.data_ptr(), hd.data_ptr(), None, 0, 0, zf.data_ptr(), c, None, rows, c, 0, st),
             "bn_apply")
    e_c = float((zc.cpu().double() - z64).abs().max())
    e_f = float((zf.cpu().double() - z64).abs().max())
    assert e_c < 2e-6 and e_f > 20 * e_c, (e_c, e_f)
    return dict(centred=e_c, fused=e_f)


def bn_small_batch_case(dev, n=4, c=32, k=64, seed=23):
    """Train-mode BatchNorm over a handful of rows (the global-average-pool branch: n x 1 x 1): the batch mean is the correctly
    rounded mean of the stored values (float64 statistics, up_bn_exact_stats_t), also when |mean| >> std.  Integer inputs and
    weights make the 1x1 convolution exact, momentum 1 makes running_mean the batch mean itself."""
    import copy
    x = torch.randint(-3, 4, (n, c, 1, 1), generator=g(seed)).float()
    conv = torch.nn.Conv2d(c, k, 1, bias=False)
    bn = torch.nn.BatchNorm2d(k, momentum=1.0)
    with torch.no_grad():
        w = torch.randint(-2, 3, tuple(conv.weight.shape), generator=g(seed + 1)).float()
        x[:, 0] = 1.0
        w[:, 0, 0, 0] = torch.randint(3000, 6000, (k,), generator=g(seed + 2)).float() + 0.25      # sums of 4 need > 24 bits
        conv.weight.copy_(w)
        bn.weight.copy_(0.5 + torch.rand(k, generator=g(seed + 3)))
        bn.bias.copy_(0.2 * torch.randn(k, generator=g(seed + 4)))
    conv_d, bn_d = copy.deepcopy(conv).to(dev), copy.deepcopy(bn).to(dev)
    conv.train(), bn.train(), conv_d.train(), bn_d.train()
    with torch.no_grad():
        yc = conv(x)
        assert torch.equal(yc.double(), torch.nn.functional.conv2d(x.double(), w.double()))
        zr = bn(yc)
    z = nchw(ops.conv_bn_act(nhwc(x, dev), conv_d, bn_d, relu=False), k)
    mean64 = yc.double().mean((0, 2, 3))
    assert torch.equal(bn_d.running_mean.cpu(), mean64.float()), (bn_d.running_mean.cpu() - mean64.float()).abs().max()
    assert torch.equal(bn.running_mean, mean64.float())                  # ATen (CPU: double accumulators) does the same
    assert rel(bn_d.running_var.cpu(), bn.running_var) < 1e-6
    # the normalised values against the float64 evaluation (ATen's CPU kernel applies x * alpha + beta' and is itself 3e-5 off here;
    # its CUDA kernel and this library evaluate the centred form)
    var64 = yc.double().var((0, 2, 3), unbiased=False)
    z64 = (yc.double() - mean64.view(1, -1, 1, 1)) / torch.sqrt(var64 + bn.eps).view(1, -1, 1, 1) * bn.weight.double().view(1, -1, 1, 1) \
        + bn.bias.double().view(1, -1, 1, 1)
    # (an fp32 mean is itself up to half an ulp off: that much, times the scale, is the floor of ANY fp32 implementation)
    bound = (0.5 * 2.0 ** -23 * mean64.abs() * bn.weight.double() / torch.sqrt(var64 + bn.eps)).view(1, -1, 1, 1) + 2e-6
    assert bool(((z.double() - z64).abs() <= bound).all()), float(((z.double() - z64).abs() / bound).max())
    assert float((zr.double() - z64).abs().max()) < 1e-3


def bn_rows_ab_case(dev, n, c, h, w, k, relu=True, residual=True, dtype=torch.float32, seed=0):
    """The row-strided BatchNorm passes (norm_act.hip: bn_apply_rows_kernel / bn_bwd_apply_rows_kernel, per-channel parameters
    loaded once per thread) against the flat ones on a conv -> BN (-> +residual) (-> ReLU) train step: same arithmetic, same
    relu_bits layout, so the forward results are bitwise equal and the gradients equal to rounding."""
    import copy
    from unipose_amd import _C
    gen = torch.Generator().manual_seed(seed)
    conv = torch.nn.Conv2d(c, k, 1, bias=False)
    bn = torch.nn.BatchNorm2d(k)
    with torch.no_grad():
        bn.weight.copy_(0.5 + torch.rand(k, generator=gen))
        bn.bias.copy_(0.2 * torch.randn(k, generator=gen))
    x0 = torch.randn(n, h, w, c, generator=gen)
    r0 = torch.randn(n, h, w, k, generator=gen)
    g0 = torch.randn(n, h, w, k, generator=gen)
    out = []
    try:
        for mode in (1, 0):
            _C.check(_C.lib().up_conv_tune(b"bn_rows", mode), "bn_rows")
            cd, bd = copy.deepcopy(conv).to(dev).train(), copy.deepcopy(bn).to(dev).train()
            x = x0.to(dtype).to(dev).requires_grad_(True)
            res = r0.to(dtype).to(dev).requires_grad_(True) if residual else None
            z = ops.conv_bn_act(x, cd, bd, relu=relu, residual=res)
            z.backward(g0.to(dtype).to(dev))
            out.append([z.detach(), x.grad, cd.weight.grad, bd.weight.grad, bd.bias.grad] + ([res.grad] if residual else []))
    finally:
        _C.lib().up_conv_tune(b"bn_rows", 1)
    # forward: bitwise.  Backward: the same formula, but with the per-channel factors loop-invariant the compiler contracts
    # the multiply-adds differently on the GPU (one fp32 ulp; one bf16 ulp on a few elements after the final rounding)
    tol = 2e-6 if dtype == torch.float32 else 1e-2
    for a, b, what in zip(out[0], out[1], ["z", "dx", "dw", "dgamma", "dbeta", "dres"]):
        a, b = a.float().cpu(), b.float().cpu()
        if what in ("z", "dres"):
            assert torch.equal(a, b), what
        else:
            err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            assert err < tol, (what, err)


def bn_groups_case(dev, groups, n, c, h, w, k, r=1, relu=True, residual=False, dtype=torch.float32, seed=0, tol=5e-5,
                   grouped_fwd=False):
    """conv -> BatchNorm inside ops.bn_groups(G) on a batch of G*n images (group-major) against G separate train-mode calls
    of torch's conv + BatchNorm on the G sub-batches with the SAME modules: outputs, data / residual gradients per group,
    parameter gradients summed over the groups, running statistics after G sequential momentum updates, the batch counter."""
    import copy
    pad = r // 2
    x = torch.randn(groups * n, c, h, w, generator=g(seed)) + 0.3
    conv = torch.nn.Conv2d(c, k, r, padding=pad, bias=False)
    bn = torch.nn.BatchNorm2d(k)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5)
        bn.weight.copy_(0.5 + torch.rand(k, generator=g(seed + 2)))
        bn.bias.copy_(0.2 * torch.randn(k, generator=g(seed + 3)))
        bn.running_mean.copy_(0.1 * torch.randn(k, generator=g(seed + 4)))
        bn.running_var.copy_(0.5 + torch.rand(k, generator=g(seed + 5)))
    bf = dtype == torch.bfloat16
    if bf:
        x = x.to(dtype).float()
        with torch.no_grad():
            conv.weight.copy_(conv.weight.to(dtype).float())
    conv_d, bn_d = copy.deepcopy(conv).to(dev).train(), copy.deepcopy(bn).to(dev).train()
    conv.train(), bn.train()
    res = torch.randn(groups * n, k, h, w, generator=g(seed + 6)) if residual else None
    if bf and residual:
        res = res.to(dtype).float()
    dy = torch.randn(groups * n, k, h, w, generator=g(seed + 7))
    if bf:
        dy = dy.to(dtype).float()
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if residual else None
    outs = []
    for gi in range(groups):                     # the reference's way: one module call per group (frame)
        sl = slice(gi * n, (gi + 1) * n)
        o = bn(conv(xr[sl]))
        if residual:
            o = o + rr[sl]
        outs.append(o)
    yr = torch.cat(outs, 0)

    def to_dev(t, ch):
        cp = ops.rup32(ch) if bf else ops.rup4(ch)
        o = torch.zeros(t.shape[0], t.shape[2], t.shape[3], cp)
        o[..., :ch] = t.permute(0, 2, 3, 1)
        return o.to(dtype).to(dev)
    xd = to_dev(x, c).requires_grad_(True)
    rd = to_dev(res, k).requires_grad_(True) if residual else None
    from unipose_amd import _C
    g0 = int(_C.lib().up_conv_counter(b"glds32_grouped"))
    prev, ops.GROUPED_TILES = ops.GROUPED_TILES, grouped_fwd        # forward statistics from per-group tiles (off by default)
    try:
        with ops.bn_groups(groups):
            y = ops.conv_bn_act(xd, conv_d, bn_d, relu=relu, residual=rd)
    finally:
        ops.GROUPED_TILES = prev
    took_grouped_tiles = int(_C.lib().up_conv_counter(b"glds32_grouped")) > g0
    yo = nchw(y, k)
    yr_fwd = F.relu(yr) if relu else yr
    if relu:
        yr = yr * (yo > 0).float()
    yr.backward(dy)
    y.backward(to_dev(dy, k))
    errs = {
        "y": rel(yo, yr_fwd.detach()),
        "dx": rel(nchw(xd.grad, c), xr.grad),
        "dw": rel(conv_d.weight.grad.cpu(), conv.weight.grad),
        "dgamma": rel(bn_d.weight.grad.cpu(), bn.weight.grad),
        "dbeta": rel(bn_d.bias.grad.cpu(), bn.bias.grad),
        "rm": rel(bn_d.running_mean.cpu(), bn.running_mean),
        "rv": rel(bn_d.running_var.cpu(), bn.running_var),
    }
    if residual:
        errs["dres"] = rel(nchw(rd.grad, k), rr.grad)
    assert int(bn_d.num_batches_tracked) == int(bn.num_batches_tracked) == groups
    # bf16 storage: the group statistics are taken from the STORED (rounded) convolution output
    bad = {k_: v for k_, v in errs.items() if not v < (tol if not bf else 2e-3 if k_ in ("rm", "rv") else 3e-2)}
    assert not bad, (bad, errs)
    errs["grouped_tiles"] = took_grouped_tiles
    return errs


def bn_group_stats_case(dev, groups, rows, c, seed=0, momentum=0.1, eps=1e-5):
    """up_bn_batch_stats_t + up_bn_finalize_groups on their own at a row count that gives the finalize kernel several rounds of
    tiles per lane (rows / 256 > 128) — the network-sized cases have 17: per-group mean / invstd / scale / shift against float64,
    running statistics after `groups` momentum updates in group order."""
    from unipose_amd import _C
    L = _C.lib()
    gen = torch.Generator().manual_seed(seed)
    y = (3.0 + 2.0 * torch.randn(groups * rows, c, generator=gen)) * (0.5 + torch.rand(c, generator=gen))
    y[rows:2 * rows] += 1.5                                      # groups differ
    gamma, beta = 0.5 + torch.rand(c, generator=gen), 0.2 * torch.randn(c, generator=gen)
    rm0, rv0 = 0.1 * torch.randn(c, generator=gen), 0.5 + torch.rand(c, generator=gen)
    yd, gd, bd = y.to(dev), gamma.to(dev), beta.to(dev)
    rm, rv = rm0.clone().to(dev), rv0.clone().to(dev)
    tiles = L.up_bn_batch_stats_tiles(rows)
    st = torch.empty((groups, tiles, c, 3), dtype=torch.float32, device=dev)
    coef = torch.empty((groups, 4, c), dtype=torch.float32, device=dev)
    _C.check(L.up_bn_batch_stats_t(yd.data_ptr(), c, rows, c, groups, 0, st.data_ptr(), ops._stream(yd)), "bn_batch_stats")
    _C.check(L.up_bn_finalize_groups(st.data_ptr(), tiles, c, groups, rows, eps, momentum, rm.data_ptr(), rv.data_ptr(),
                                     gd.data_ptr(), bd.data_ptr(), coef.data_ptr(), ops._stream(yd)), "bn_finalize_groups")
    y64 = y.double().view(groups, rows, c)
    mean, var = y64.mean(1), y64.var(1, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + eps)
    co = coef.cpu().double()
    assert float((co[:, 0] - mean).abs().max()) < 2e-6 * float(mean.abs().max())
    assert float((co[:, 1] / invstd - 1).abs().max()) < 5e-6
    assert float((co[:, 2] / (gamma.double() * invstd) - 1).abs().max()) < 5e-6
    assert float((co[:, 3] - (beta.double() - mean * gamma.double() * invstd)).abs().max()) < 2e-5
    erm, erv = rm0.double(), rv0.double()
    for g in range(groups):
        erm = (1 - momentum) * erm + momentum * mean[g]
        erv = (1 - momentum) * erv + momentum * var[g] * rows / (rows - 1)
    assert float((rm.cpu().double() - erm).abs().max()) < 2e-6
    assert float((rv.cpu().double() / erv - 1).abs().max()) < 5e-6


def bn_groups_chain_case(dev, groups, n, c, h, w, k1, k2, r2=1, tol=2e-5, seed=0):
    """conv1 -> BN -> ReLU -> conv2 -> BN -> ReLU inside ops.bn_groups(G), the first BatchNorm's backward reduction riding in conv2's
    data gradient (ops.BnSlot with row groups: that launch is tiled per group and reduces every group's sums against the group's own
    mean / invstd) against the same chain with the separate grouped reduce pass (itself pinned to G torch calls by bn_groups_case)."""
    from unipose_amd import _C
    torch.manual_seed(seed)
    conv1, bn1 = torch.nn.Conv2d(c, k1, 1, bias=False), torch.nn.BatchNorm2d(k1)
    conv2, bn2 = torch.nn.Conv2d(k1, k2, r2, padding=r2 // 2, bias=False), torch.nn.BatchNorm2d(k2)
    with torch.no_grad():
        for b in (bn1, bn2):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
    mods = [m.to(dev).train() for m in (conv1, bn1, conv2, bn2)]
    x0 = torch.randn(groups * n, h, w, c) + torch.arange(groups).repeat_interleave(n).view(-1, 1, 1, 1) * 0.5   # groups differ
    g0 = torch.randn(groups * n, h, w, k2)
    out = {}
    prev = ops.GROUPED_REDUCE
    try:
        for fused in (True, False):
            ops.GROUPED_REDUCE = fused
            for m in mods:
                m.zero_grad(set_to_none=True)
            x = x0.clone().to(dev).requires_grad_(True)
            u0, b0 = ops.HOST_COUNTERS["bn_prereduced"], int(_C.lib().up_conv_counter(b"glds32_bnred"))
            f0 = ops.HOST_COUNTERS["bn_bwd_folded"]
            q0 = int(_C.lib().up_conv_counter(b"glds32_grouped"))
            with ops.bn_groups(groups):
                s1 = ops.BnSlot()
                y1 = ops.conv_bn_act(x, mods[0], mods[1], relu=True, slot_out=s1)
                y2 = ops.conv_bn_act(y1, mods[2], mods[3], relu=True, slot_in=s1)
            (y2 * g0.to(dev)).sum().backward()
            ops.wgrad_fence()
            used = ops.HOST_COUNTERS["bn_prereduced"] - u0
            assert used == (1 if fused else 0), used
            merged = ops.HOST_COUNTERS["bn_bwd_folded"] - f0
            assert merged == (1 if fused else 0), merged     # round 6: that launch also merged every group's sums (bn_fold.h)
            assert (int(_C.lib().up_conv_counter(b"glds32_bnred")) - b0) == (1 if fused else 0)
            assert (int(_C.lib().up_conv_counter(b"glds32_grouped")) - q0) == (1 if fused else 0)
            out[fused] = {"y2": y2.detach().cpu(), "dx": x.grad.cpu(),
                          **{f"p{i}": p.grad.cpu() for i, m in enumerate(mods) for p in m.parameters()}}
    finally:
        ops.GROUPED_REDUCE = prev
    worst = max((rel(out[True][n_], out[False][n_]), n_) for n_ in out[True])
    assert worst[0] < tol, worst
    return worst


def optimizer_stale_case(dev, math, fused, nweights=3):
    """torch's fused optimizers (one multi-tensor kernel) update the parameters WITHOUT moving their version counters — the
    packed weight images the convolutions read must still follow (ops.invalidate_packed_weights, called from a global
    optimizer post-step hook).  Found by the G16 trajectory golden in round 5: with Adam(fused=True) every convolution kept
    running on the weights of step 0.  Both image caches (fp32, bf16 planes), both re-packed by their batched launch; with more
    than 24 parameters on a GPU the launch is split between the current and the side stream (ops._launch_repack)."""
    from unipose_amd import ops
    gen = torch.Generator().manual_seed(3)
    ws = [torch.nn.Parameter((torch.randn(32, 32, 3, 3, generator=gen) * 0.05).to(dev)) for _ in range(nweights)]
    x = torch.randn(2, 6, 6, 32, generator=gen).to(dev)
    cfg = ops.ConvCfg(1, 1, 1)
    opt = torch.optim.Adam(ws, lr=0.05, fused=fused)
    tol = 1e-5 if math == "f32" else 3e-2
    other = torch.nn.Parameter((torch.randn(32, 32, 3, 3, generator=gen) * 0.05).to(dev))     # a second "model": not this optimizer's
    ops.set_conv_math(math)
    try:
        y_other = ops.conv_fwd_raw(x, other, cfg)[0].float().cpu()
        cache = (ops._PACK_CACHE if math == "f32" else ops._PACK16_CACHE)[other.device.index]
        for step in range(3):
            order = range(nweights) if step % 2 == 0 else reversed(range(nweights))     # (the side-stream half first, too)
            for i in order:
                y = ops.conv_fwd_raw(x, ws[i], cfg)[0].float().cpu()
                r = torch.nn.functional.conv2d(x.cpu().permute(0, 3, 1, 2), ws[i].detach().cpu(), padding=1).permute(0, 2, 3, 1)
                err = float((y - r).abs().max() / r.abs().max())
                assert err < tol, (step, i, err)
            v0 = [w._version for w in ws]
            for w in ws:
                w.grad = torch.randn(w.shape, generator=gen).to(dev)
            opt.step()
            if fused:
                assert [w._version for w in ws] == v0, "torch's fused Adam now moves the version counter: the hook is belt and braces"
            assert cache.entries[id(other)][1] == other._version, "a parameter of no stepping optimizer keeps its images"
            assert all(cache.entries[id(w)][1] == -1 for w in ws)
        assert torch.equal(ops.conv_fwd_raw(x, other, cfg)[0].float().cpu(), y_other)       # (a sub-table re-pack left it alone)
    finally:
        ops.set_conv_math("f32")


def bn_fold_case(dev, n, c, h, w, k1, k2, r2=1, dtype=torch.float32, cus=0, seed=0):
    """conv1 -> BN -> ReLU -> conv2 -> BN -> ReLU, one training step, with the BatchNorm finalize FOLDED into the producing launches
    (forward: the convolution's last workgroup per channel column merges the statistics partials, up_bn_fold; backward: the data
    gradient that carries the reduction also merges it, up_bn_reduce_slot.dgamma / folded; bn2's reduce pass takes the tickets in
    bn_bwd_reduce_kernel) against the same step with up_conv_tune("bn_fold", 0), where the stand-alone arrive kernel runs the
    same merge tree: EQUAL bits everywhere, and the host counters prove which form ran.
    cus: shrink the "chip" so that the launches have K-split tail tiles (only their finishing workgroup arrives)."""
    from unipose_amd import _C
    torch.manual_seed(seed)
    conv1, bn1 = torch.nn.Conv2d(c, k1, 1, bias=False), torch.nn.BatchNorm2d(k1)
    conv2, bn2 = torch.nn.Conv2d(k1, k2, r2, padding=r2 // 2, bias=False), torch.nn.BatchNorm2d(k2)
    with torch.no_grad():
        for b in (bn1, bn2):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
            b.running_mean.normal_(0, 0.1)
    mods = [m.to(dev).train() for m in (conv1, bn1, conv2, bn2)]
    state0 = [{k_: v.clone() for k_, v in m.state_dict().items()} for m in mods]
    x0 = (torch.randn(n, h, w, c) + 0.3).to(dtype)
    g0 = torch.randn(n, h, w, k2).to(dtype)
    L = _C.lib()
    out, counts = {}, {}
    if cus:
        _C.check(L.up_conv_tune(b"cu_count", cus), "tune")
    try:
        for fold in (1, 0):
            _C.check(L.up_conv_tune(b"bn_fold", fold), "tune")
            for m, s0 in zip(mods, state0):
                m.load_state_dict(s0)
                m.zero_grad(set_to_none=True)
            x = x0.clone().to(dev).requires_grad_(True)
            f0, b0 = ops.HOST_COUNTERS["bn_fwd_folded"], ops.HOST_COUNTERS["bn_bwd_folded"]
            s1 = ops.BnSlot()
            y1 = ops.conv_bn_act(x, mods[0], mods[1], relu=True, slot_out=s1)
            y2 = ops.conv_bn_act(y1, mods[2], mods[3], relu=True, slot_in=s1)
            (y2.float() * g0.to(dev).float()).sum().backward()
            ops.wgrad_fence()
            counts[fold] = (ops.HOST_COUNTERS["bn_fwd_folded"] - f0, ops.HOST_COUNTERS["bn_bwd_folded"] - b0)
            out[fold] = {"y2": y2.detach().float().cpu(), "dx": x.grad.float().cpu(),
                         **{f"p{i}.{nm}": p.grad.cpu() for i, m in enumerate(mods) for nm, p in m.named_parameters()},
                         **{f"b{i}.{nm}": b.detach().clone().cpu() for i, m in enumerate(mods) for nm, b in m.named_buffers()}}
    finally:
        _C.check(L.up_conv_tune(b"bn_fold", 1), "tune")
        if cus:
            _C.check(L.up_conv_tune(b"cu_count", 0), "tune")
    assert counts[0] == (0, 0), counts
    assert counts[1][0] == 2, counts                  # both convolutions finalized their own statistics
    if dtype == torch.float32 and c % 32 == 0 and k1 % 32 == 0:
        assert counts[1][1] == 1, counts              # conv2's data gradient reduced AND merged bn1's sums
    for k_ in out[1]:
        assert torch.equal(out[1][k_], out[0][k_]), (k_, float((out[1][k_].double() - out[0][k_].double()).abs().max()))
    # ... and the statistics themselves are right: batch mean / biased variance of conv1's output against float64
    with torch.no_grad():
        y = torch.nn.functional.conv2d(x0.double().permute(0, 3, 1, 2), state0[0]["weight"].double().cpu())
        mean, var = y.mean((0, 2, 3)), y.var((0, 2, 3), unbiased=True)
        rm = 0.9 * state0[1]["running_mean"].double().cpu() + 0.1 * mean
        rv = 0.9 * state0[1]["running_var"].double().cpu() + 0.1 * var
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel(out[1]["b1.running_mean"].double(), rm) < tol, rel(out[1]["b1.running_mean"].double(), rm)
    assert rel(out[1]["b1.running_var"].double(), rv) < tol, rel(out[1]["b1.running_var"].double(), rv)
    return counts[1]


def bn_groups_fold_case(dev, groups, n, c, h, w, k1, k2, r2=1, seed=0):
    """Row groups (ops.bn_groups) with the finalize folded into the producers — the statistics pass merges every group's partial
    rows (up_bn_stats_groups_t), the grouped data gradient / reduce pass merges the backward sums per group and over the groups
    (up_bn_reduce_slot.gsum) — against up_conv_tune("bn_fold", 0), where the stand-alone arrive kernels run the same merge tree:
    EQUAL bits for outputs, gradients and running statistics."""
    from unipose_amd import _C
    torch.manual_seed(seed)
    conv1, bn1 = torch.nn.Conv2d(c, k1, 1, bias=False), torch.nn.BatchNorm2d(k1)
    conv2, bn2 = torch.nn.Conv2d(k1, k2, r2, padding=r2 // 2, bias=False), torch.nn.BatchNorm2d(k2)
    with torch.no_grad():
        for b in (bn1, bn2):
            b.weight.uniform_(0.5, 1.5)
            b.bias.normal_(0, 0.3)
    mods = [m.to(dev).train() for m in (conv1, bn1, conv2, bn2)]
    state0 = [{k_: v.clone() for k_, v in m.state_dict().items()} for m in mods]
    x0 = torch.randn(groups * n, h, w, c) + torch.arange(groups).repeat_interleave(n).view(-1, 1, 1, 1) * 0.5
    g0 = torch.randn(groups * n, h, w, k2)
    L = _C.lib()
    out, merged = {}, {}
    try:
        for fold in (1, 0):
            _C.check(L.up_conv_tune(b"bn_fold", fold), "tune")
            for m, s0 in zip(mods, state0):
                m.load_state_dict(s0)
                m.zero_grad(set_to_none=True)
            x = x0.clone().to(dev).requires_grad_(True)
            f0 = ops.HOST_COUNTERS["bn_bwd_folded"]
            with ops.bn_groups(groups):
                s1 = ops.BnSlot()
                y1 = ops.conv_bn_act(x, mods[0], mods[1], relu=True, slot_out=s1)
                y2 = ops.conv_bn_act(y1, mods[2], mods[3], relu=True, slot_in=s1)
            (y2 * g0.to(dev)).sum().backward()
            ops.wgrad_fence()
            merged[fold] = ops.HOST_COUNTERS["bn_bwd_folded"] - f0
            out[fold] = {"y2": y2.detach().cpu(), "dx": x.grad.cpu(),
                         **{f"p{i}.{nm}": p.grad.cpu() for i, m in enumerate(mods) for nm, p in m.named_parameters()},
                         **{f"b{i}.{nm}": b.detach().clone().cpu() for i, m in enumerate(mods) for nm, b in m.named_buffers()}}
    finally:
        _C.check(L.up_conv_tune(b"bn_fold", 1), "tune")
    assert merged == {1: 1, 0: 0}, merged
    for k_ in out[1]:
        assert torch.equal(out[1][k_], out[0][k_]), (k_, float((out[1][k_].double() - out[0][k_].double()).abs().max()))
    return merged


def stem_ab_case(dev, n, size, seed=0):
    """The first convolution (7x7, stride 2, padding 3, 3 -> 64 channels, resnet.py:113) on stem7_kernel (stem_f32.h: input patch in
    LDS, weights in registers) against igemm_kernel<128,64,generic>: same MFMA, same k order -> EQUAL outputs; the BatchNorm partials
    (same tiles, same epilogue) too.  size >= 256 (an output row of at least 128 pixels) or the launch stays on the generic kernel."""
    from unipose_amd import _C
    L = _C.lib()
    x = nhwc(torch.randn(n, 3, size, size, generator=g(seed)), dev)
    w = (torch.randn(64, 3, 7, 7, generator=g(seed + 1)) * 0.05).to(dev)
    cfg = ops.ConvCfg(2, 3, 1)
    out = {}
    try:
        _C.check(L.up_conv_tune(b"tile_want", 1), "tile_want")      # (small batches: keep the 128 x 64 tiles the network's launch takes)
        for mode in (1, 0):
            _C.check(L.up_conv_tune(b"stem7", mode), "stem7")
            c0 = L.up_conv_counter(b"stem7")
            y, d, st = ops.conv_fwd_raw(x, w, cfg, stats=True)
            assert L.up_conv_counter(b"stem7") - c0 == mode, "launches on stem7_kernel"
            out[mode] = (y.cpu(), st.cpu())
    finally:
        L.up_conv_tune(b"stem7", 0)
        L.up_conv_tune(b"tile_want", 1500)
    (y1, s1), (y0, s0) = out[1], out[0]
    assert torch.equal(y1, y0), ("stem output", float((y1 - y0).abs().max()))
    assert torch.equal(s1, s0), ("stem BatchNorm partials", float((s1 - s0).abs().max()))
    ref = F.conv2d(x.cpu()[..., :3].permute(0, 3, 1, 2), w.cpu(), stride=2, padding=3)
    err = rel(nchw(y1, 64), ref)
    assert err < 2e-5, err
    return y1
