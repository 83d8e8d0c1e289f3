"""In-process A/B of run-time knobs (up_conv_tune) on the BASELINE configs[1] training step:
the model is built once, variants are timed interleaved.  A variant is 'name=value+name=value' ('base' = defaults).
    python tools/gpu/tune_ab.py --rounds 3 --steps 5 base glds32=0+tail_split=0 tile_want=1000"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

DEFAULTS = {"tile_want": 1500, "tile_want_bf16": 500, "tail_split": 1, "split_per_cu": 2, "tap_sort": 1, "wgrad_rect": 1,
            "glds": 1, "glds32": 1, "glds32_epi": 1, "glds32_wgrad": 1, "bn_rows": 1, "tiny_k": 128}
# (round 2, profiles/r02_a_knob_ab.txt: occ64, wgrad_single and the high-priority main stream measured no gain and were removed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--eval-only", action="store_true", help="time the inference forward instead of the train step")
    args = ap.parse_args()
    from model.unipose import unipose
    from unipose_amd import _C, ops
    dev = torch.device("cuda:0")
    ops._side_stream(dev)
    lib = _C.lib()
    K, B, S = 16, args.batch, 368
    torch.manual_seed(0)
    model = unipose("MPII", num_classes=K).to(dev).train()
    x = torch.randn(B, 3, S, S).to(dev)
    t = torch.rand(B, K + 1, S // 8, S // 8).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)

    def apply(spec):
        kv = dict(DEFAULTS)
        if spec != "base":
            for item in spec.split("+"):
                k, v = item.split("=")
                kv[k] = int(v)
        for k, v in kv.items():
            _C.check(lib.up_conv_tune(k.encode(), v), "tune " + k)

    def step():
        if args.eval_only:
            with torch.no_grad():
                return model(x)
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

    if args.eval_only:
        model.eval()
    for v in args.variants:
        apply(v)
        timed(2)
    res = {v: [] for v in args.variants}
    for r in range(args.rounds):
        for v in args.variants:
            apply(v)
            timed(1)
            res[v].append(round(timed(args.steps), 3))
        print("round", r, {k: q[-1] for k, q in res.items()}, flush=True)
    apply("base")
    print(json.dumps({"ms_per_step": res, "best": {k: min(q) for k, q in res.items()}}))


if __name__ == "__main__":
    main()
