"""What the weight-gradient launches of a real training step look like (workgroups per launch), next to the plan the
library reports for the same layer: guards against a knob silently differing between a bare process and a torch one."""
import collections, csv, ctypes, os, sys
sys.path.insert(0, os.getcwd())
os.environ["UP_PROFILE_CSV"] = "/tmp/q_live.csv"
import torch
from model.unipose import unipose
from unipose_amd import _C, ops
lib = _C.lib()
dev = torch.device("cuda:0")
m = unipose("MPII", num_classes=16).to(dev).train()
x = torch.randn(32, 3, 368, 368, device=dev); t = torch.rand(32, 17, 46, 46, device=dev)
# both stream modes: the per-launch CSV of round 1 taken with UNIPOSE_SYNC_WGRAD=1 (weight gradients on the main stream)
# showed 252 / 256 workgroups per layer3 launch, this script in the default two-stream mode 504 / 480 — same planner, same
# descriptors; if the two modes still differ, bench.py's `exclusive` pass times another split plan than the timed region
for mode in ("two streams", "one stream"):
    ops.ASYNC_WGRAD = mode == "two streams"
    for rep in range(2):
        if rep == 1:
            lib.up_profile_begin()
        m.zero_grad(set_to_none=True)
        loss = ops.mse_loss(m(x), t); loss.backward(); torch.cuda.synchronize()
    arr = (ctypes.c_double * (lib.up_profile_variants() * 3))(); lib.up_profile_end(arr, lib.up_profile_variants())
    c = collections.Counter()
    for r in csv.DictReader(open("/tmp/q_live.csv")):
        if r["kernel"].startswith("wgrad"):
            c[(r["kernel"], r["M"], r["N"], r["K"], r["workgroups"])] += 1
    print(mode)
    for k, n in c.most_common(4):
        print("  ", n, k)
print("multiProcessorCount", torch.cuda.get_device_properties(0).multi_processor_count)
