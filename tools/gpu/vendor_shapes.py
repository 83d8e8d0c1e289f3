"""The step's dominant convolution shapes on the vendor path (PyTorch-ROCm -> MIOpen / hipBLASLt, fp32) next to this
library's kernels, forward / data gradient / weight gradient, each timed alone with HIP events:
    python tools/gpu/vendor_shapes.py            (prints one table; a few seconds per shape once MIOpen has its kernels)
A yardstick for the MFMA fractions in DESIGN.md: what the stock fp32 convolution of this platform reaches on the same
problem.  Not part of the product path and not used by any test."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# (name, N, C, H, W, K, R, stride, pad, dil) — SURVEY T1 rows that carry the step (B = 32, 368^2 input)
SHAPES = [
    ("layer3 3x3 256->256 @23", 32, 256, 23, 23, 256, 3, 1, 1, 1),
    ("layer3 1x1 256->1024 @23", 32, 256, 23, 23, 1024, 1, 1, 0, 1),
    ("layer3 1x1 1024->256 @23", 32, 1024, 23, 23, 256, 1, 1, 0, 1),
    ("layer2 3x3 128->128 @46", 32, 128, 46, 46, 128, 3, 1, 1, 1),
    ("layer4 3x3 512->512 d2 @23", 32, 512, 23, 23, 512, 3, 1, 2, 2),
    ("layer1 3x3 64->64 @92", 32, 64, 92, 92, 64, 3, 1, 1, 1),
    ("WASP 3x3 256->256 d12 @23", 32, 256, 23, 23, 256, 3, 1, 12, 12),
]


def ev_time(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    from unipose_amd import ops
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    budget = float(os.environ.get("VENDOR_BUDGET_S", "200"))
    t_start = time.time()
    print(f"{'shape':30s} {'pass':6s} {'vendor ms':>10s} {'TF':>7s} {'vendor cl ms':>13s} {'TF':>7s} {'this ms':>9s} {'TF':>7s}")
    for name, n, c, h, w, k, r, st, pad, dil in SHAPES:
        if time.time() - t_start > budget:
            print("(time budget reached)")
            break
        g = torch.Generator(device="cpu").manual_seed(0)
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        wt = (torch.randn(k, c, r, r, generator=g) * 0.05).to(dev)
        p = (h + 2 * pad - dil * (r - 1) - 1) // st + 1
        flop = 2.0 * n * p * p * k * c * r * r
        dy = torch.randn(n, k, p, p, generator=g).to(dev)
        x_cl, dy_cl, w_cl = (t.contiguous(memory_format=torch.channels_last) for t in (x, dy, wt))
        # this library: NHWC tensors through the same entry points the model uses
        xh = x.permute(0, 2, 3, 1).contiguous()
        dyh = dy.permute(0, 2, 3, 1).contiguous()
        cfg = ops.ConvCfg(st, pad, dil)
        yh, d, _ = ops.conv_fwd_raw(xh, wt, cfg)
        rows = {}
        rows["fwd"] = (lambda: F.conv2d(x, wt, None, st, pad, dil), lambda: F.conv2d(x_cl, w_cl, None, st, pad, dil),
                       lambda: ops.conv_fwd_raw(xh, wt, cfg, out=yh))
        bwd = torch.ops.aten.convolution_backward
        args = ([st, st], [pad, pad], [dil, dil], False, [0, 0], 1)
        rows["dgrad"] = (lambda: bwd(dy, x, wt, None, *args, [True, False, False]),
                         lambda: bwd(dy_cl, x_cl, w_cl, None, *args, [True, False, False]),
                         lambda: ops.conv_bwd_data_raw(dyh, wt, d, xh.shape, dev))
        rows["wgrad"] = (lambda: bwd(dy, x, wt, None, *args, [False, True, False]),
                         lambda: bwd(dy_cl, x_cl, w_cl, None, *args, [False, True, False]),
                         lambda: ops.conv_bwd_weight_raw(xh, dyh, wt.shape, d, False))
        for pas, (f_v, f_cl, f_me) in rows.items():
            try:
                tv = ev_time(f_v)
            except Exception as ex:   # noqa: BLE001
                tv = float("nan")
                print("   vendor NCHW failed:", str(ex)[:80])
            try:
                tc = ev_time(f_cl)
            except Exception as ex:   # noqa: BLE001
                tc = float("nan")
                print("   vendor channels_last failed:", str(ex)[:80])
            tm = ev_time(f_me)
            print(f"{name:30s} {pas:6s} {tv:10.4f} {flop / tv / 1e9:7.1f} {tc:13.4f} {flop / tc / 1e9:7.1f} {tm:9.4f} {flop / tm / 1e9:7.1f}",
                  flush=True)


if __name__ == "__main__":
    main()
