"""bf16-storage forward convolutions of the 46x46 stage (BASELINE configs[4]) as a function of the batch, i.e. of the number of
tiles per CU, each launch alone on the GPU (library HIP events), under values of up_conv_tune knobs:
    python tools/gpu/bf16_quant.py base tile_want_bf16=600 tile_want_bf16=2200
Prints us / TFLOP/s / algorithmic GB/s per (shape, batch, variant): is the launch time a step function of ceil(tiles / CUs)
(per-CU bound: tile quantisation matters) or linear in the tiles (chip-level bound)?"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

# (label, c, hw, k, r, pad, dil)
SHAPES = [
    ("3x3 256->256 @46", 256, 46, 256, 3, 1, 1),
    ("1x1 1024->256 @46", 1024, 46, 256, 1, 0, 1),
    ("1x1 256->1024 @46", 256, 46, 1024, 1, 0, 1),
    ("1x1 64->256 @184", 64, 184, 256, 1, 0, 1),
    ("1x1 128->512 @92", 128, 92, 512, 1, 0, 1),
    ("3x3 512->512 d2 @46", 512, 46, 512, 3, 2, 2),
]
BATCHES = {46: [7, 8, 15, 16, 23, 24, 32, 48, 64], 184: [4, 8, 16], 92: [4, 8, 16, 32]}


def main():
    from unipose_amd import _C, ops
    variants = sys.argv[1:] or ["base"]
    dev = torch.device("cuda:0")
    lib = _C.lib()
    nv = lib.up_profile_variants()
    defaults = {}

    def apply(spec):
        for k, v in defaults.items():
            _C.check(lib.up_conv_tune(k.encode(), v), "tune " + k)
        if spec != "base":
            for item in spec.split("+"):
                k, v = item.split("=")
                _C.check(lib.up_conv_tune(k.encode(), int(v)), "tune " + k)

    for spec in variants:
        if spec != "base":
            for item in spec.split("+"):
                defaults.setdefault(item.split("=")[0], {"tile_want_bf16": 500, "big_stages": 3, "glds_big": 1, "big_min_k": 256}.get(item.split("=")[0], 0))

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
    for label, c, hw, k, r, pad, dil in SHAPES:
        w = (torch.randn(k, c, r, r, generator=g) * 0.02).to(dev)
        cfg = ops.ConvCfg(1, pad, dil)
        for n in BATCHES[hw]:
            x = torch.randn(n, hw, hw, c, generator=g).to(torch.bfloat16).to(dev)
            y, _, _ = ops.conv_fwd_raw(x, w, cfg, stats=True)
            flop = 2.0 * n * hw * hw * k * c * r * r
            byts = 2.0 * n * hw * hw * (c + k) + 2.0 * k * c * r * r
            best = {v: 1e9 for v in variants}
            for rnd in range(2):
                for v in variants:
                    apply(v)
                    timed(x, w, cfg, y, 3)
                    best[v] = min(best[v], timed(x, w, cfg, y, 15))
            m = n * hw * hw
            print(f"{label:22s} B={n:3d} M={m:7d} tiles128={-(-m // 128) * -(-k // 128):5d}  " +
                  "  ".join(f"{v}: {best[v] * 1e3:7.1f} us {flop / best[v] / 1e9:6.0f} TF {byts / best[v] / 1e6:5.0f} GB/s" for v in variants),
                  flush=True)
            del x, y
    apply("base")


if __name__ == "__main__":
    main()
