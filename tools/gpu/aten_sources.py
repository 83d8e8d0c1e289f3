"""Which lines of this repository issue the FOREIGN (non-up::) kernels of a training step: one step of bench.make_workload under
torch.profiler (CPU activity, Python stacks), ATen operators that launch device work grouped by the innermost frame inside the repository.
    python tools/gpu/aten_sources.py [--model lstm] [--batch 8] [--size 368]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

# operators that launch device work themselves; one that only wraps another of the list (zero_ -> fill_, clone -> copy_) is not counted
LEAVES = ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::copy_", "aten::cat", "aten::mul", "aten::mul_", "aten::sum",
          "aten::where", "aten::index_select", "aten::_foreach_add_", "aten::div", "aten::sub", "aten::neg", "aten::constant_pad_nd",
          "aten::_fused_adam_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::_to_copy", "aten::to")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="lstm")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=368)
    ap.add_argument("--frames", type=int, default=5)
    args = ap.parse_args()
    import bench
    dev = torch.device("cuda:0")
    lstm = args.model == "lstm"
    model, opt, step = bench.make_workload(dev, lstm, 13 if lstm else 16, args.batch, args.size, args.frames, 0)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as p:
        step()
        torch.cuda.synchronize()
    total = collections.Counter()
    where = collections.Counter()
    for e in p.events():
        if e.name in LEAVES and not any(c.name in LEAVES for c in (e.cpu_children or [])):
            total[e.name] += 1
            frames = [s for s in (e.stack or []) if ROOT in s or "unipose_amd" in s or "bench.py" in s]
            frames = [s for s in frames if "aten_sources" not in s]
            where[(e.name, frames[0].strip() if frames else "(outside the repository: %s)" % ((e.stack or ["?"])[0].strip()))] += 1
    print("operators that launch device work, one step:", dict(total))
    for (name, frame), c in where.most_common(60):
        print(f"{c:5d}  {name:22s} {frame}")


if __name__ == "__main__":
    main()
