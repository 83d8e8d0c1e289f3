"""Which host-side torch ops issue the small fill / copy kernels of a training step?  torch.profiler over one step of the
headline workload, aten::fill_ / aten::zero_ / aten::copy_ / aten::zeros grouped by Python stack."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
model, opt, step = bench.make_workload(dev, False, 16, int(os.environ.get("B", "8")), 368, 5, 0)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous",
                  "aten::add", "aten::add_", "aten::cat"):
        st = [s for s in e.stack if "unipose_amd" in s or "bench.py" in s or "model/" in s][:3]
        cnt[(e.name, " <- ".join(s.split("/")[-1] for s in st))] += 1
for (name, st), n in cnt.most_common(40):
    print(f"{n:5d}  {name:18s} {st}")
