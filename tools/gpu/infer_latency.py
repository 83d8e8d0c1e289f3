"""Inference forward latency of the image model at 368x368, eager launches vs ONE hipGraph (unipose_amd/graph.py), fp32 and
bf16 storage:  python tools/gpu/infer_latency.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402


def wall(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from model.unipose import unipose
    from unipose_amd import ops
    from unipose_amd.graph import GraphedForward
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unipose("MPII", num_classes=16).to(dev).eval()
    print(f"{'math':6s} {'B':>3s} {'eager ms':>9s} {'graph ms':>9s} {'img/s (graph)':>14s}")
    for math in ("f32", "bf16s"):
        ops.set_conv_math(math)
        for B in (1, 2, 4, 8, 16, 32):
            x = torch.randn(B, 3, 368, 368, device=dev)

            def eager():
                with torch.no_grad():
                    model(x)

            fwd = GraphedForward(model, x)
            te, tg = wall(eager, 30), wall(lambda: fwd(x), 30)
            print(f"{math:6s} {B:3d} {te:9.3f} {tg:9.3f} {B * 1e3 / tg:14.1f}", flush=True)
            del fwd
    ops.set_conv_math("f32")


if __name__ == "__main__":
    main()
