"""Is the host ahead of the GPU?  Host clock and stream events at the phase boundaries of the bench step (zero_grad | forward |
backward | optimizer), NO synchronisation between steps: per phase the host issue time, the GPU time between the events, and the
host's lead over the GPU at each boundary (negative lead = the GPU waited for the host there).
    python tools/gpu/skew.py [--size 368 --batch 32 --math f32 --steps 12]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=368)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--math", default="f32")
    ap.add_argument("--steps", type=int, default=12)
    args = ap.parse_args()
    import bench
    from unipose_amd import ops
    dev = torch.device("cuda:0")
    ops._side_stream(dev)
    ops.set_conv_math(args.math)
    model, opt, _ = bench.make_workload(dev, False, 16, args.batch, args.size, 1, seed=0)
    x = torch.randn(args.batch, 3, args.size, args.size, device=dev)
    t = torch.rand(args.batch, 17, args.size // 8, args.size // 8, device=dev)
    names = ["(nothing)", "zero_grad", "forward", "backward", "optimizer"]

    def one(rec):
        def mark():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            rec.append((time.perf_counter(), e))
        mark()
        mark()              # two consecutive records: what an empty interval reads
        opt.zero_grad(set_to_none=True)
        mark()
        loss = ops.mse_loss(model(x), t)
        mark()
        loss.backward()
        mark()
        opt.step()
        mark()

    for _ in range(4):
        one([])
    torch.cuda.synchronize()
    base_e = torch.cuda.Event(enable_timing=True)
    base_e.record()
    torch.cuda.synchronize()
    base_h = time.perf_counter()
    recs = []
    for _ in range(args.steps):
        r = []
        one(r)
        recs.append(r)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - base_h) * 1e3 / args.steps
    print(f"wall per step {wall:.2f} ms (unsynchronised loop of {args.steps} steps)")
    print(f"{'step':>4} " + " ".join(f"{n + ' host/gpu':>22}" for n in names) + f" {'lead at: fwd end':>18} {'bwd end':>9} {'opt end':>9}")
    tot_h = [0.0] * 5
    tot_g = [0.0] * 5
    for i, r in enumerate(recs):
        h = [(a - base_h) * 1e3 for a, _ in r]
        g = [base_e.elapsed_time(e) for _, e in r]
        cells = []
        for k in range(5):
            tot_h[k] += h[k + 1] - h[k]
            tot_g[k] += g[k + 1] - g[k]
            cells.append(f"{h[k + 1] - h[k]:9.2f} /{g[k + 1] - g[k]:9.2f}  ")
        lead = [g[k] - h[k] for k in (3, 4, 5)]
        print(f"{i:4d} " + " ".join(cells) + f" {lead[0]:18.2f} {lead[1]:9.2f} {lead[2]:9.2f}")
    n = len(recs)
    print("mean " + " ".join(f"{tot_h[k] / n:9.2f} /{tot_g[k] / n:9.2f}  " for k in range(5)))
    print(f"host issue time per step {sum(tot_h) / n:.2f} ms; GPU time per step {sum(tot_g) / n:.2f} ms")


if __name__ == "__main__":
    main()
