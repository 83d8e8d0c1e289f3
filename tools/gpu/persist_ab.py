"""A/B of the persistent stream-K convolution launches against the default form, in ONE process (model built once):
parity of an eval forward and of one train step's loss, then interleaved timings of the BASELINE configs[1] step.
    python tools/gpu/persist_ab.py [--rounds 3] [--steps 5] [--grids 0,512,1024]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--grids", default="0")
    ap.add_argument("--variants", default="", help="extra persistent variants 'tpw:xcd:grid,...' (e.g. 0:0:0,100:1:0)")
    ap.add_argument("--csv", default="", help="directory: per-launch timings (one step, both streams serialised) per variant")
    args = ap.parse_args()
    from model.unipose import unipose
    from unipose_amd import _C, ops
    dev = torch.device("cuda:0")
    ops._side_stream(dev)
    lib = _C.lib()
    K, B, S = 16, args.batch, 368
    torch.manual_seed(0)
    model = unipose("MPII", num_classes=K).to(dev)
    x = torch.randn(B, 3, S, S).to(dev)
    t = torch.rand(B, K + 1, S // 8, S // 8).to(dev)
    out = {}

    def mode(on, grid=0, tpw=100, xcd=0):
        _C.check(lib.up_conv_set_persistent(on, grid), "set_persistent")
        _C.check(lib.up_conv_tune(b"persist_tpw", tpw), "tune")
        _C.check(lib.up_conv_tune(b"persist_xcd", xcd), "tune")

    # parity: eval forward (deterministic) default vs persistent
    model.eval()
    with torch.no_grad():
        mode(0)
        y0 = model(x[:4]).float().cpu()
        mode(1)
        y1 = model(x[:4]).float().cpu()
    out["eval_max_rel"] = float((y1 - y0).abs().max() / y0.abs().max())
    out["argmax_equal"] = bool(torch.equal(y0.flatten(2).argmax(2), y1.flatten(2).argmax(2)))
    print("parity", out, flush=True)

    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = ops.mse_loss(model(x), t)
        loss.backward()
        opt.step()
        return loss

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    variants = [("default", 0, 0, 100, 0)]
    if args.grids:
        variants += [(f"persistent(grid={g})", 1, int(g), 100, 0) for g in args.grids.split(",")]
    for v in filter(None, args.variants.split(",")):
        tpw, xcd, grid = (int(q) for q in v.split(":"))
        variants.append((f"persistent(tpw={tpw},xcd={xcd},grid={grid})", 1, grid, tpw, xcd))
    for name, on, grid, tpw, xcd in variants:          # warm every variant (scratch, pack buffers, allocator)
        mode(on, grid, tpw, xcd)
        timed(2)
    if args.csv:
        import ctypes
        os.makedirs(args.csv, exist_ok=True)
        nv = lib.up_profile_variants()
        for i, (name, on, grid, tpw, xcd) in enumerate(variants):
            mode(on, grid, tpw, xcd)
            os.environ["UP_PROFILE_CSV"] = os.path.join(args.csv, f"v{i}.csv")
            torch.cuda.synchronize()
            lib.up_profile_begin()
            step()
            torch.cuda.synchronize()
            arr = (ctypes.c_double * (nv * 3))()
            lib.up_profile_end(arr, nv)
            with open(os.path.join(args.csv, f"v{i}.name"), "w") as f:
                f.write(name)
        os.environ.pop("UP_PROFILE_CSV", None)
    res = {name: [] for name, *_ in variants}
    for r in range(args.rounds):
        for name, on, grid, tpw, xcd in variants:
            mode(on, grid, tpw, xcd)
            timed(1)
            res[name].append(round(timed(args.steps), 3))
        print("round", r, {k: v[-1] for k, v in res.items()}, flush=True)
    mode(0)
    out["ms_per_step"] = res
    out["best"] = {k: min(v) for k, v in res.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
