"""Stress of the K-split tail tiles of the direct-to-LDS bf16 kernels: the layer3 3x3 / 1x1 launches of the 736^2 step, split on
against split off, many repetitions with other launches in between; reports where (tile, rows) a mismatch beyond re-association
round-off sits."""
import sys
import torch
import os
sys.path.insert(0, ".")
from unipose_amd import _C, ops

if os.environ.get("UP_LIB"):          # A/B of two builds: UP_LIB=tools/gpu/ab/lib_old.so
    _C.load(os.environ["UP_LIB"])
    print("library:", os.environ["UP_LIB"])

dev = torch.device("cuda")
BF = torch.bfloat16


def tune(**kw):
    for k, v in kw.items():
        _C.check(_C.lib().up_conv_tune(k.encode(), int(v)), k)


def run(c, k, r, pad, kt, reps, q=4, maxp=256, sync=0):
    g = torch.Generator().manual_seed(c + k + r)
    x = torch.randn(16, 46, 46, c, generator=g).to(BF).to(dev)
    wt = (torch.randn(k, c, r, r, generator=g) * (2.0 / (c * r * r)) ** 0.5).to(dev)
    cfg = ops.ConvCfg(1, pad, 1)
    tune(tile_want_bf16=500, glds_kt=kt, glds=1, glds_split=0)
    y0 = ops.conv_fwd_raw(x, wt, cfg, stats=True)[0].float()
    bad = 0
    for i in range(reps):
        tune(glds_split=1, glds_split_q=q, glds_split_maxp=maxp)
        y1 = ops.conv_fwd_raw(x, wt, cfg, stats=True)[0].float()
        if i % 3 == 0:      # another split launch of a different shape in between (other flags / shares contents)
            ops.conv_fwd_raw(x[:, :, :, :c // 2].contiguous() if c >= 64 else x, wt[:, :c // 2].contiguous() if c >= 64 else wt, cfg)
        err = (y1 - y0).abs()
        big = err > 0.05 * y0.abs().clamp_min(0.5)
        if bool(big.any()):
            bad += 1
            if bad > 3:
                continue
            idx = big.nonzero()
            rows = (idx[:, 0] * 46 * 46 + idx[:, 1] * 46 + idx[:, 2])
            print(f"  rep {i}: {int(big.sum())} elements off, max {float(err.max()):.3f}, pixels {int(rows.min())}..{int(rows.max())}, "
                  f"channels {int(idx[:, 3].min())}..{int(idx[:, 3].max())}")
    print(f"c={c} k={k} r={r} kt={kt} q={q} maxp={maxp} sync={sync}: {bad} of {reps} repetitions wrong")
    tune(glds_split=0, glds_split_q=2, glds_split_maxp=4, glds_kt=32)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for kt in (64, 32):
        run(256, 256, 3, 1, kt, reps)
        run(1024, 256, 1, 0, kt, reps)
        run(256, 256, 3, 1, kt, reps, q=2, maxp=4)
