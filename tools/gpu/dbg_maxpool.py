import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bf16s_cases as bc
from unipose_amd import _C, ops
dev = torch.device("cuda:0")
L = _C.lib()
x = torch.randn(2, 16, 9, 10, generator=bc.g(3))
xb = bc.rb(x)
x32 = xb.permute(0, 2, 3, 1).contiguous().to(dev)
x16 = x32.to(torch.bfloat16)
n, h, w, c = x32.shape
p, q = 5, 5
st = torch.cuda.current_stream().cuda_stream
out = {}
for name, xin, dti, dto in (("ff", x32, 0, 0), ("tt", x16, 1, 1), ("tf", x16, 1, 0), ("ft", x32, 0, 1)):
    y = torch.empty((n, p, q, c), dtype=torch.bfloat16 if dto else torch.float32, device=dev)
    idx = torch.full((n, p, q, c), 77, dtype=torch.uint8, device=dev)
    _C.check(L.up_maxpool3s2_fwd_t(xin.data_ptr(), c, y.data_ptr(), c, idx.data_ptr(), n, h, w, c, p, q, dti, dto, st), name)
    torch.cuda.synchronize()
    out[name] = (y.float().cpu(), idx.cpu())
for name in ("tt", "tf", "ft"):
    dv = int((out[name][0] != out["ff"][0]).sum()); di = (out[name][1] != out["ff"][1])
    print(name, "value mismatches", dv, "idx mismatches", int(di.sum()))
    for b in di.nonzero()[:4]:
        nn, pp, qq, cc = [int(v) for v in b]
        win = xb[nn, cc, max(2 * pp - 1, 0):2 * pp + 2, max(2 * qq - 1, 0):2 * qq + 2]
        print("   at", (nn, pp, qq, cc), "idx", int(out[name][1][nn, pp, qq, cc]), "vs", int(out["ff"][1][nn, pp, qq, cc]), "window", win.tolist())
