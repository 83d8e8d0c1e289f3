"""Small-batch latency of the image model (inference forward at B = 1 / 4 / 8 and the B = 8 train step) under values of a
run-time knob:  python tools/gpu/small_batch.py split_per_cu 1 2 3 4"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402


def main():
    key, values = sys.argv[1], [int(v) for v in sys.argv[2:]]
    from model.unipose import unipose
    from unipose_amd import _C, ops
    dev = torch.device("cuda:0")
    ops._side_stream(dev)
    lib = _C.lib()
    torch.manual_seed(0)
    model = unipose("MPII", num_classes=16).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    rows = {}
    for B in (1, 4, 8):
        x = torch.randn(B, 3, 368, 368).to(dev)
        t = torch.rand(B, 17, 46, 46).to(dev)

        def fwd():
            with torch.no_grad():
                return model(x)

        def train():
            opt.zero_grad(set_to_none=True)
            ops.mse_loss(model(x), t).backward()
            opt.step()

        for mode, fn, n in (("eval", fwd, 20), ("train", train, 8)):
            if mode == "train" and B == 1:
                continue
            model.train(mode == "train")
            for r in range(2):
                for v in values:
                    _C.check(lib.up_conv_tune(key.encode(), v), key)
                    timed(fn, 2)
                    rows.setdefault((mode, B, v), []).append(round(timed(fn, n), 3))
    for (mode, B, v), ts in rows.items():
        print(f"{mode:5s} B={B} {key}={v}: {min(ts):8.3f} ms  {ts}")


if __name__ == "__main__":
    main()
