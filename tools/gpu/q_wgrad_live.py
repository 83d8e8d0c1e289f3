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
for rep in range(2):
    if rep == 1:
        lib.up_profile_begin()
    loss = ops.mse_loss(m(x), t); loss.backward(); torch.cuda.synchronize()
arr = (ctypes.c_double * (lib.up_profile_variants() * 3))(); lib.up_profile_end(arr, lib.up_profile_variants())
c = collections.Counter()
for r in csv.DictReader(open("/tmp/q_live.csv")):
    if r["kernel"].startswith("wgrad"):
        c[(r["kernel"], r["M"], r["N"], r["K"], r["workgroups"])] += 1
for k, n in c.most_common(6):
    print(n, k)
print("multiProcessorCount", torch.cuda.get_device_properties(0).multi_processor_count)
