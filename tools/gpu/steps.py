"""Per-step wall times of one bench workload (every step synchronised): tells a host-bound or allocator-bound step from a
kernel-bound one.  python tools/gpu/steps.py --size 736 --batch 16 --math bf16 --steps 8"""
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
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--sync-wgrad", action="store_true")
    ap.add_argument("--lstm", action="store_true", help="UniPose-LSTM (K=13, --frames frames) instead of the image model")
    ap.add_argument("--frames", type=int, default=5)
    args = ap.parse_args()
    import bench
    from unipose_amd import ops
    dev = torch.device("cuda:0")
    ops._side_stream(dev)
    ops.set_conv_math(args.math)
    if args.sync_wgrad:
        ops.ASYNC_WGRAD = False
    model, opt, step = bench.make_workload(dev, args.lstm, 13 if args.lstm else 16, args.batch, args.size,
                                           args.frames if args.lstm else 1, seed=0)
    ts = []
    for i in range(args.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3, 1), round((t2 - t0) * 1e3, 1)))
    print("per step (host ms, total ms):", ts, "reserved GB", round(torch.cuda.memory_reserved() / 2 ** 30, 1))


if __name__ == "__main__":
    main()
