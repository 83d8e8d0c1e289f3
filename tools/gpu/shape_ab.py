"""Per-shape A/B of run-time knobs (up_conv_tune) on the forward convolutions that lose the most time against the MFMA peak
(profiles/r04_z_lost_time_by_shape.txt), each launch alone on the GPU, timed with the library's per-launch HIP events:
    python tools/gpu/shape_ab.py base tail_per_cu=4 persist=1+stagger=4 ...
One process, variants interleaved per shape; prints ms and TFLOP/s per (shape, variant) and the sum weighted by launches per step."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

# (label, n, c, hw, k, r, pad, dil, launches per step as forward OR data gradient of the same GEMM shape)
SHAPES = [
    ("3x3 256->256 @92 (K2304, 16928 tiles)", 32, 256, 92, 256, 3, 1, 1, 0),     # many dispatch waves: the K loop without quantisation
    ("1x1 1024->256 @92 (K1024, 16928 tiles)", 32, 1024, 92, 256, 1, 0, 1, 0),
    ("1x1 256->1024 @46 (K256, 16928 tiles)", 32, 256, 46, 1024, 1, 0, 1, 0),
    ("3x3 256->256 @23 (K2304 N256)", 32, 256, 23, 256, 3, 1, 1, 51),
    ("1x1 1024->256 @23 (K1024 N256)", 32, 1024, 23, 256, 1, 0, 1, 46),
    ("1x1 256->1024 @23 (K256 N1024)", 32, 256, 23, 1024, 1, 0, 1, 45),
    ("1x1 64->256 @92 (K64 N256)", 32, 64, 92, 256, 1, 0, 1, 6),
    ("1x1 512->2048 @23 (K512 N2048)", 32, 512, 23, 2048, 1, 0, 1, 5),
    ("1x1 128->512 @46 (K128 N512)", 32, 128, 46, 512, 1, 0, 1, 7),
    ("3x3 64->64 @92 (K576 N64)", 32, 64, 92, 64, 3, 1, 1, 6),
    ("3x3 128->128 @46 (K1152 N128)", 32, 128, 46, 128, 3, 1, 1, 7),
    ("1x1 2048->512 @23 (K2048 N512)", 32, 2048, 23, 512, 1, 0, 1, 5),
    ("1x1 256->64 @92 (K256 N64)", 32, 256, 92, 64, 1, 0, 1, 6),
    ("3x3 512->512 d2 @23 (K4608 N512)", 32, 512, 23, 512, 3, 2, 2, 2),
    ("3x3 256->256 d18 @23 (WASP)", 32, 256, 23, 256, 3, 18, 18, 2),
]
DEFAULTS = {"tiny_k": 128, "tail_split": 1, "tile_want": 1500}


def main():
    from unipose_amd import _C, ops
    variants = sys.argv[1:] or ["base"]
    dev = torch.device("cuda:0")
    lib = _C.lib()
    nv = lib.up_profile_variants()

    def apply(spec):
        kv = dict(DEFAULTS)
        if spec != "base":
            for item in spec.split("+"):
                k, v = item.split("=")
                kv[k] = int(v)
        for k, v in kv.items():
            _C.check(lib.up_conv_tune(k.encode(), v), "tune " + k)

    def timed(x, w, cfg, y, iters):
        arr = (ctypes.c_double * (nv * 3))()
        lib.up_profile_begin()
        for _ in range(iters):
            ops.conv_fwd_raw(x, w, cfg, out=y, stats=True)
        torch.cuda.synchronize(dev)
        _C.check(lib.up_profile_end(arr, nv), "profile_end")
        n = sum(arr[i * 3] for i in range(nv))
        return sum(arr[i * 3 + 1] for i in range(nv)) / max(n, 1.0)

    g = torch.Generator().manual_seed(1)
    total = {v: 0.0 for v in variants}
    table = []
    for label, n, c, hw, k, r, pad, dil, per_step in SHAPES:
        x = torch.randn(n, hw, hw, c, generator=g).to(dev)
        w = (torch.randn(k, c, r, r, generator=g) * 0.02).to(dev)
        cfg = ops.ConvCfg(1, pad, dil)
        y, _, _ = ops.conv_fwd_raw(x, w, cfg, stats=True)
        flop = 2.0 * n * hw * hw * k * c * r * r
        best = {v: 1e9 for v in variants}
        for rnd in range(3):
            for v in variants:
                apply(v)
                timed(x, w, cfg, y, 3)
                best[v] = min(best[v], timed(x, w, cfg, y, 20))
        row = {"shape": label, "per_step": per_step}
        for v in variants:
            row[v] = {"us": round(best[v] * 1e3, 2), "tf": round(flop / best[v] / 1e9, 1)}
            total[v] += best[v] * per_step
        table.append(row)
        print(f"{label:36s} x{per_step:3d}  " + "  ".join(f"{v}: {row[v]['us']:8.2f} us {row[v]['tf']:6.1f} TF" for v in variants), flush=True)
        del x, w, y
    apply("base")
    print("weighted ms per step:", {v: round(t, 3) for v, t in total.items()})
    print(json.dumps({"table": table, "weighted_ms_per_step": total}))


if __name__ == "__main__":
    main()
