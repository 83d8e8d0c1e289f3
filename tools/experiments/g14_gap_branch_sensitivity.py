"""How far do the G14 gradients of the genuine reference move when its convolutions are summed in another order?
Every Conv2d output of the reference (CPU) is perturbed by sigma x one fp32 ulp of random round-off — what another summation order
of the same dot products does — and the relative L2 distance of each gradient from the unperturbed run is printed.
Measured (build container): sigma = 1: wasp.conv2 1.9e-3 ... 2.6e-3, wasp.aspp2 3.5e-3 ... 4.3e-3, decoder.conv1 3.3e-3 ... 4.1e-3 (two
seeds); sigma = 2: 2.2e-3 / 4.6e-3 / 3.8e-3.  The fixture's own yardsticks (fp32 vs fp64 of the same kernels; an exact-statistics
BatchNorm behind the same convolution outputs) read 1.6e-3 / 2.7e-3 / 2.7e-3 for these keys: they do not move the inputs of the
BatchNorm behind the global-average-pool branch (wasp.py:51-53), which normalises B = 4 values per channel with |mean| / std = 364
on this input — an ulp of its input is 2e-5 of the normalised scale.  `--gap-only` perturbs only that branch's convolution.
Run: python tools/experiments/g14_gap_branch_sensitivity.py [--gap-only]   (needs /root/reference: build container only)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import make_goldens as G
from oracle import unipose_oracle as O

K, B, size = 16, 4, 368
x = O.synth_input((B, 3, size, size), 61)
t = O.synth_input((B, K + 1, size // 8, size // 8), 62, "rand")
KEYS = ["backbone.layer4.2.conv2.weight", "wasp.conv2.weight", "wasp.aspp2.atrous_conv.weight", "wasp.global_avg_pool.1.weight",
        "decoder.conv1.weight", "decoder.last_conv.0.weight"]


def run(ulps, seed=0, gap_only=False):
    m = G.ref_image_model(K, 9).train()
    m.wasp.dropout.p = 0.0
    m.decoder.last_conv[3].p = 0.0
    m.decoder.last_conv[7].p = 0.0
    if ulps:
        mods = [m.wasp.global_avg_pool[1]] if gap_only else [q for q in m.modules() if isinstance(q, torch.nn.Conv2d)]
        for i, mod in enumerate(mods):
            def hook(_m, _i, o, i=i):
                gen = torch.Generator().manual_seed(1000 * seed + i)
                r = torch.randn(o.shape, generator=gen).to(o.dtype)
                ulp = torch.pow(2.0, torch.floor(torch.log2(o.detach().abs().clamp_min(1e-30))) - 23)
                return o + r * ulps * ulp
            mod.register_forward_hook(hook)
    y = m(x)
    torch.nn.MSELoss()(y, t).backward()
    return dict(m.named_parameters())


if __name__ == "__main__":
    gap_only = "--gap-only" in sys.argv
    ref = run(0)
    for u, sd in ((1, 0), (1, 1), (2, 0)):
        g = run(u, sd, gap_only)
        print(f"sigma {u} ulp, seed {sd}:", "  ".join(
            f"{k.split('.')[0]}.{k.split('.')[1]} "
            f"{float((g[k].grad.double() - ref[k].grad.double()).norm() / ref[k].grad.double().norm()):.2e}" for k in KEYS))
