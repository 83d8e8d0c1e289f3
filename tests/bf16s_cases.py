"""bf16-STORAGE parity cases (BASELINE configs[4]) shared by the emulator and the GPU tests.  Inputs are rounded to bf16
first, so the fp32 torch reference sees exactly the operands the kernels see; what remains is the bf16 rounding of the
OUTPUT (2^-9 relative) and, for convolutions, of nothing else (bf16 x bf16 products are exact in fp32)."""
import torch
import torch.nn.functional as F

from unipose_amd import ops

BF = torch.bfloat16


def g(seed):
    gen = torch.Generator()
    gen.manual_seed(seed)
    return gen


def rb(t):
    """round to bf16 and back"""
    return t.to(BF).float()


def nhwc16(x, dev, pad_to=None):
    """NCHW fp32 cpu tensor -> NHWC bf16 on dev, channels padded with zeros to a multiple of 8 (or pad_to)."""
    n, c, h, w = x.shape
    cp = pad_to or (c + 7) // 8 * 8
    y = torch.zeros(n, h, w, cp)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.to(BF).to(dev)


def nchw(y, c):
    return y.detach().float().cpu()[..., :c].permute(0, 3, 1, 2).contiguous()


def rel(a, b):
    a, b = a.double(), b.double()
    s = b.abs().max().item()
    return (a - b).abs().max().item() / (s if s > 0 else 1.0)


OUT_TOL = 6e-3      # one bf16 rounding of the result: 2^-9 = 2e-3 of the element, measured against the tensor maximum


def conv_case(dev, n, c, h, w, k, r, stride, pad, dil, bias=False, seed=0):
    """ConvBias in bf16 storage: forward, data gradient, weight gradient (+ bias gradient)."""
    x = rb(torch.randn(n, c, h, w, generator=g(seed)))
    wt = torch.randn(k, c, r, r, generator=g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5
    b = torch.randn(k, generator=g(seed + 2)) if bias else None
    xd = nhwc16(x, dev, pad_to=(c + 31) // 32 * 32).requires_grad_(True)
    wd = wt.clone().to(dev).requires_grad_(True)
    bd = b.clone().to(dev).requires_grad_(True) if bias else None
    y = ops.ConvBias.apply(xd, wd, bd, ops.ConvCfg(stride, pad, dil), False)
    assert y.dtype == BF and y.shape[3] == ops.rup32(k)
    xr = x.clone().requires_grad_(True)
    wr = rb(wt).requires_grad_(True)                       # the kernels round the weights to bf16 as well
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
    dy = rb(torch.randn(yr.shape, generator=g(seed + 3)))
    yr.backward(dy)
    y.backward(nhwc16(dy, dev, pad_to=ops.rup32(k)))
    errs = {"y": rel(nchw(y, k), yr.detach()), "dx": rel(nchw(xd.grad, c), xr.grad), "dw": rel(wd.grad.cpu(), wr.grad)}
    if bias:
        errs["db"] = rel(bd.grad.cpu(), br.grad)
    if ops.rup32(k) != k:
        assert float(y.detach().float()[..., k:].abs().max()) == 0.0
    assert float(xd.grad.float()[..., c:].abs().max()) == 0.0 if xd.shape[3] != c else True
    assert wd.grad.dtype == torch.float32
    bad = {k_: v for k_, v in errs.items() if not v < (OUT_TOL if k_ != "dw" and k_ != "db" else 1e-4)}
    assert not bad, (bad, errs)
    return errs


def conv_bn_case(dev, n, c, h, w, k, r, stride, pad, dil, relu=True, residual=False, train=True, seed=0):
    """conv -> BatchNorm (batch statistics from the fp32 accumulators) -> (+residual) -> (ReLU) with bf16 tensors; the
    reference rounds the convolution output to bf16 where the kernel does (y is stored, then normalised)."""
    import copy
    x = rb(torch.randn(n, c, h, w, generator=g(seed)) + 0.3)
    conv = torch.nn.Conv2d(c, k, r, stride=stride, padding=pad, dilation=dil, bias=False)
    bn = torch.nn.BatchNorm2d(k)
    with torch.no_grad():
        conv.weight.copy_(rb(torch.randn(conv.weight.shape, generator=g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5))
        bn.weight.copy_(0.5 + torch.rand(k, generator=g(seed + 2)))
        bn.bias.copy_(0.2 * torch.randn(k, generator=g(seed + 3)))
    conv_d, bn_d = copy.deepcopy(conv).to(dev), copy.deepcopy(bn).to(dev)
    conv.train(train), bn.train(train), conv_d.train(train), bn_d.train(train)
    xr = x.clone().requires_grad_(True)
    yr = bn(conv(xr))
    res = rb(torch.randn(yr.shape, generator=g(seed + 6))) if residual else None
    rr = res.clone().requires_grad_(True) if residual else None
    if residual:
        yr = yr + rr
    xd = nhwc16(x, dev, pad_to=(c + 31) // 32 * 32).requires_grad_(True)
    rd = nhwc16(res, dev, pad_to=ops.rup32(k)).requires_grad_(True) if residual else None
    y = ops.conv_bn_act(xd, conv_d, bn_d, relu=relu, residual=rd)
    assert y.dtype == BF
    yr_fwd = F.relu(yr) if relu else yr
    if relu:
        yr = yr * (nchw(y, k) > 0).float()
    dy = rb(torch.randn(yr.shape, generator=g(seed + 7)))
    yr.backward(dy)
    y.backward(nhwc16(dy, dev, pad_to=ops.rup32(k)))
    errs = {
        "y": rel(nchw(y, k), yr_fwd.detach()),
        "dx": rel(nchw(xd.grad, c), xr.grad),
        "dw": rel(conv_d.weight.grad.cpu(), conv.weight.grad),
        "dgamma": rel(bn_d.weight.grad.cpu(), bn.weight.grad),
        "dbeta": rel(bn_d.bias.grad.cpu(), bn.bias.grad),
        "rm": rel(bn_d.running_mean.cpu(), bn.running_mean),
        "rv": rel(bn_d.running_var.cpu(), bn.running_var),
    }
    if residual:
        errs["dres"] = rel(nchw(rd.grad, k), rr.grad)
    # y (the raw convolution output) is rounded to bf16 before it is normalised, dy of the BatchNorm likewise: every
    # quantity downstream carries a few 2^-9 roundings
    bad = {k_: v for k_, v in errs.items() if not v < (3e-2 if k_ in ("dx", "dw", "dgamma", "dbeta", "dres") else 1.5e-2)}
    assert not bad, (bad, errs)
    if ops.rup32(k) != k:
        assert float(y.detach().float()[..., k:].abs().max()) == 0.0
    return errs


def small_ops_case(dev):
    # fp32 -> bf16 max-pool (the stem boundary) and its bf16 -> fp32 backward
    x = torch.randn(2, 16, 9, 10, generator=g(3))
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = rb(torch.randn(yr.shape, generator=g(4)))
    yr.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    y = ops.MaxPool3s2.apply(xd, BF)
    assert y.dtype == BF
    y.backward(nhwc16(dy, dev))
    assert torch.equal(nchw(y, 16), rb(yr.detach())) and xd.grad.dtype == torch.float32
    assert rel(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad) < 1e-6
    # bf16 -> bf16 max-pool
    xb = rb(x)
    xr = xb.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    yr.backward(dy)
    xd = nhwc16(xb, dev).requires_grad_(True)
    y = ops.MaxPool3s2.apply(xd)
    y.backward(nhwc16(dy, dev))
    assert torch.equal(nchw(y, 16), yr.detach())
    gd, gr_ = nchw(xd.grad, 16), rb(xr.grad)
    assert torch.equal(gd, gr_), (float((gd - gr_).abs().max()), int((gd != gr_).sum()), gd[gd != gr_][:4], gr_[gd != gr_][:4],
                                  xr.grad[gd != gr_][:4])
    # bilinear
    xb = rb(torch.randn(2, 8, 5, 7, generator=g(5)))
    xr = xb.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(9, 12), mode="bilinear", align_corners=True)
    dy = rb(torch.randn(yr.shape, generator=g(6)))
    yr.backward(dy)
    xd = nhwc16(xb, dev).requires_grad_(True)
    y = ops.Bilinear.apply(xd, 9, 12)
    y.backward(nhwc16(dy, dev))
    assert rel(nchw(y, 8), yr.detach()) < OUT_TOL and rel(nchw(xd.grad, 8), xr.grad) < OUT_TOL
    # global average pool
    xb = rb(torch.randn(3, 72, 5, 7, generator=g(7)))
    xr = xb.clone().requires_grad_(True)
    yr = F.adaptive_avg_pool2d(xr, 1)
    dy = rb(torch.randn(yr.shape, generator=g(8)))
    yr.backward(dy)
    xd = nhwc16(xb, dev).requires_grad_(True)
    y = ops.GlobalAvgPool.apply(xd)
    y.backward(nhwc16(dy, dev))
    assert rel(nchw(y, 72), yr.detach()) < OUT_TOL and rel(nchw(xd.grad, 72), xr.grad) < OUT_TOL
    # concat (bit-exact copy) + slices back
    a, b = rb(torch.randn(2, 16, 3, 4, generator=g(9))), rb(torch.randn(2, 8, 3, 4, generator=g(10)))
    ad, bd = nhwc16(a, dev).requires_grad_(True), nhwc16(b, dev).requires_grad_(True)
    y = ops.ConcatC.apply(32, ad, bd)
    assert y.shape[3] == 32 and torch.equal(nchw(y, 24), torch.cat([a, b], 1)) and float(y.detach().float()[..., 24:].abs().max()) == 0
    dy = rb(torch.randn(2, 32, 3, 4, generator=g(11)))
    y.backward(nhwc16(dy, dev))
    assert torch.equal(nchw(ad.grad, 16), dy[:, :16]) and torch.equal(nchw(bd.grad, 8), dy[:, 16:24])
    # dropout with an injected mask
    xb = rb(torch.randn(2, 8, 3, 4, generator=g(12)))
    mk = (torch.rand(2, 3, 4, 8, generator=g(13)) > 0.5).float()
    xd = nhwc16(xb, dev).requires_grad_(True)
    y = ops.Dropout.apply(xd, 0.5, 1, mk.to(dev))
    assert torch.equal(y.detach().float().cpu(), rb(xb.permute(0, 2, 3, 1) * mk * 2.0))
    y.backward(torch.ones_like(y))
    assert torch.equal(xd.grad.float().cpu(), mk * 2.0)
    # NHWC bf16 -> NCHW fp32 (the network's output) and back
    xb = rb(torch.randn(2, 17, 5, 6, generator=g(14)))
    xd = nhwc16(xb, dev, pad_to=32).requires_grad_(True)
    y = ops.ToNCHW.apply(xd, 17)
    assert y.dtype == torch.float32 and torch.equal(y.detach().cpu(), xb)
    dy = torch.randn(2, 17, 5, 6, generator=g(15))
    y.backward(dy.to(dev))
    assert xd.grad.dtype == BF and torch.equal(nchw(xd.grad, 17), rb(dy)) and float(xd.grad.float()[..., 17:].abs().max()) == 0


def f32_out_case(dev, n=2, c=64, h=9, w=9, k=17, r=1, pad=0, seed=3):
    """The network's last convolution in bf16 storage writes fp32 (UP_MATH_BF16S_F32OUT): same accumulators as the bf16-output
    launch (rounding the fp32 result gives the bf16 result exactly), fp32 round-off from torch, gradients flow back as bf16."""
    x = rb(torch.randn(n, c, h, w, generator=g(seed)))
    wt = torch.randn(k, c, r, r, generator=g(seed + 1)) * (2.0 / (c * r * r)) ** 0.5
    b = torch.randn(k, generator=g(seed + 2))
    cfg = ops.ConvCfg(1, pad, 1)
    xd = nhwc16(x, dev, pad_to=(c + 31) // 32 * 32).requires_grad_(True)
    wd = wt.clone().to(dev).requires_grad_(True)
    bd = b.clone().to(dev).requires_grad_(True)
    y32 = ops.ConvBias.apply(xd, wd, bd, cfg, False, True)
    y16 = ops.ConvBias.apply(xd.detach(), wd.detach(), bd.detach(), cfg, False)
    assert y32.dtype == torch.float32 and y16.dtype == BF and y32.shape == y16.shape
    assert torch.equal(y32.detach().to(BF).cpu(), y16.cpu())
    assert float(y32.detach()[..., k:].abs().max()) == 0          # pad channels
    xr = x.clone().requires_grad_(True)
    wr = rb(wt).requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=pad)
    assert rel(nchw(y32, k), yr.detach()) < 2e-6                  # fp32 accumulation of exact bf16 products
    dy = torch.randn(yr.shape, generator=g(seed + 3))
    yr.backward(rb(dy))                                           # the gradient is rounded to bf16 on its way back
    dyd = torch.zeros(y32.shape)
    dyd[..., :k] = dy.permute(0, 2, 3, 1)
    y32.backward(dyd.to(dev))
    assert xd.grad.dtype == BF
    assert rel(nchw(xd.grad, c), xr.grad) < OUT_TOL
    assert rel(wd.grad.cpu(), wr.grad) < 1e-5 and rel(bd.grad.cpu(), br.grad) < 1e-5
    # fp32 tensors: the flag changes nothing
    xf = torch.zeros(n, h, w, (c + 31) // 32 * 32)
    xf[..., :c] = x.permute(0, 2, 3, 1)
    a_ = ops.ConvBias.apply(xf.to(dev), wd.detach(), bd.detach(), cfg, False, True)
    b_ = ops.ConvBias.apply(xf.to(dev), wd.detach(), bd.detach(), cfg, False, False)
    assert a_.dtype == torch.float32 and torch.equal(a_.cpu(), b_.cpu())


def model_eval_case(dev, K=14, B=1, size=64, tol=5e-2):
    """Whole network in bf16 storage against the fp32 oracle (own tolerance, SURVEY 8d: <= 5e-2 of the map maximum)."""
    import model_cases as mc
    from oracle import unipose_oracle as O
    m, sd = mc.build_image_model(K, 1, dev)
    m.eval()
    x = O.synth_input((B, 3, size, size), 5)
    ops.set_conv_math("bf16s")
    try:
        with torch.no_grad():
            y = m(x.to(dev))
    finally:
        ops.set_conv_math("f32")
    with torch.no_grad():
        yr = O.unipose_forward(sd, x)
    assert y.dtype == torch.float32 and y.shape == yr.shape
    e = O.max_rel(y.cpu(), yr)
    assert e < tol, e
    return e


def model_train_case(dev, K=16, B=4, size=64, cos_min=0.5, cos_head=0.9):
    """One train step in bf16 storage against the fp32 oracle: loss within 2 %, running statistics within 2 %, every trained
    parameter gets a finite fp32 gradient of the right size (norm within 25 %) and direction.  How close the direction can
    be is a property of the PROBLEM, not of the kernels: BatchNorm over a handful of samples amplifies the 2^-9 roundings
    (at B = 2 the existing bf16-operand mode with fp32 storage lands at cosine 0.25 in the trunk on this input, and the
    global-pool branch, which normalises B values per channel, has an exactly-zero true input gradient that any rounding
    replaces by noise).  The defaults are for B = 4 at 64x64 (4x4 top maps): >= 0.5 everywhere, >= 0.9 at the head."""
    import model_cases as mc
    from oracle import unipose_oracle as O
    m, sd = mc.build_image_model(K, 3, dev)
    m.train()
    for d in (m.wasp.dropout, m.decoder.last_conv[3], m.decoder.last_conv[7]):
        d.p = 0.0
    x = O.synth_input((B, 3, size, size), 13)
    t = O.synth_input((B, K + 1, size // 8, size // 8), 14, "rand")
    ops.set_conv_math("bf16s")
    try:
        y = m(x.to(dev))
        loss = ops.mse_loss(y, t.to(dev))
        loss.backward()
    finally:
        ops.set_conv_math("f32")
    sdr = O.clone_sd(sd, requires_grad=True)
    lr = F.mse_loss(O.unipose_forward(sdr, x, train=True, p_drop=(0, 0, 0)), t)
    lr.backward()
    assert abs(float(loss.detach()) - float(lr.detach())) < 2e-2 * abs(float(lr.detach())), (float(loss.detach()), float(lr.detach()))
    cos, ratio = {}, {}
    for name, p in m.named_parameters():
        gr = sdr[name].grad
        if gr is None:
            assert p.grad is None, name
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), name
        if p.grad.numel() >= 4096:
            a, b = p.grad.cpu().double().flatten(), gr.double().flatten()
            cos[name] = float(a @ b / (a.norm() * b.norm() + 1e-300))
            ratio[name] = float(a.norm() / (b.norm() + 1e-300))
    worst = min(cos, key=cos.get)
    print("bf16-storage train step: gradient cosines vs the fp32 oracle, worst five:",
          [(k_, round(v, 3)) for k_, v in sorted(cos.items(), key=lambda kv: kv[1])[:5]],
          "head:", round(cos["decoder.last_conv.8.weight"], 4), round(cos["decoder.last_conv.4.weight"], 4))
    assert cos[worst] > cos_min, (worst, cos[worst])
    assert cos["decoder.last_conv.8.weight"] > 0.99 and cos["decoder.last_conv.4.weight"] > cos_head
    off = {k_: v for k_, v in ratio.items() if not 0.8 < v < 1.25}
    assert not off, off
    msd = m.state_dict()
    for k, v in sdr.items():
        if "running_" in k and not k.startswith("decoder.bn2"):
            assert O.max_rel(msd[k].cpu(), v) < 2e-2, k
    return cos[worst]


def model_train_yardstick_case(dev, golden_file):
    """The bf16-storage train step against G13: the genuine reference run (a) under bf16 autocast and (b) in fp32 arithmetic
    with every stored tensor rounded to bf16 once — what a CORRECT bf16 implementation gets on this input (K=16, B=8,
    128x128, residual branches damped so that the comparison is not drowned in ReLU flips).  Per parameter (>= 4096
    elements) the HIP path's gradient must be at most twice as far from the fp32 gradient as the worse of the two
    yardsticks, in direction (1 - cosine) and in relative L2; the fp32 gradient is the oracle's (pinned to the reference by
    G4 / G11)."""
    import numpy as np
    import model_cases as mc
    from oracle import unipose_oracle as O
    g = np.load(golden_file)
    K, wseed, xseed, tseed, B = (int(v) for v in g["meta"])
    m, sd = mc.build_image_model(K, wseed, dev)
    sd = {k: (v * 0.25 if k.endswith("bn3.weight") else v) for k, v in sd.items()}
    m.load_state_dict(sd)
    m.train()
    for d in (m.wasp.dropout, m.decoder.last_conv[3], m.decoder.last_conv[7]):
        d.p = 0.0
    x = O.synth_input((B, 3, 128, 128), xseed)
    t = O.synth_input((B, K + 1, 16, 16), tseed, "rand")
    ops.set_conv_math("bf16s")
    try:
        y = m(x.to(dev))
        loss = ops.mse_loss(y, t.to(dev))
        loss.backward()
        ops.wgrad_fence()
    finally:
        ops.set_conv_math("f32")
    sdr = O.clone_sd(sd, requires_grad=True)
    lr = F.mse_loss(O.unipose_forward(sdr, x, train=True, p_drop=(0, 0, 0)), t)
    lr.backward()
    assert abs(float(lr.detach()) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))      # same problem as the fixture
    assert abs(float(loss.detach()) - float(lr.detach())) < 2.0 * max(abs(float(g["autocast_loss"]) - float(g["loss"])),
                                                                      abs(float(g["rounded_loss"]) - float(g["loss"]))) + 1e-3
    params = dict(m.named_parameters())
    worst = {}
    for i, name in enumerate(g["names"]):
        name = str(name)
        a, b = params[name].grad.cpu().double().flatten(), sdr[name].grad.double().flatten()
        cos = float(a @ b / (a.norm() * b.norm() + 1e-300))
        rel = float((a - b).norm() / (b.norm() + 1e-300))
        ycos = min(float(g["autocast_cos"][i]), float(g["rounded_cos"][i]))
        yrel = max(float(g["autocast_rel"][i]), float(g["rounded_rel"][i]))
        if 1.0 - cos > 2.0 * (1.0 - ycos) + 0.01 or rel > 2.0 * yrel + 0.02:
            worst[name] = (cos, ycos, rel, yrel)
    ratios = [(1.0 - float(params[str(n)].grad.cpu().double().flatten() @ sdr[str(n)].grad.double().flatten() /
                     (params[str(n)].grad.cpu().double().norm() * sdr[str(n)].grad.double().norm() + 1e-300))) /
              (1.0 - min(float(g["autocast_cos"][i]), float(g["rounded_cos"][i]))) for i, n in enumerate(g["names"])]
    print("bf16-storage gradients: (1 - cos) relative to the bf16 yardstick of the reference, median %.2f max %.2f" %
          (float(np.median(ratios)), max(ratios)))
    assert not worst, worst
    return max(ratios)
