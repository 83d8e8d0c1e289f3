"""Where do the workgroups of a CU-masked stream run?  (round 6: UNIPOSE_SIDE_CUS experiment)
    python tools/gpu/cu_mask.py 64 [stride]     lowest 64 mask bits (or every stride-th bit): histogram of XCC ids / CUs hit"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from unipose_amd import _C, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
bits, b = [], 0
while len(bits) < n:
    if b not in bits:
        bits.append(b)
    b = (b + stride) % 256
    if b in bits:
        b = (b + 1) % 256
for name, st in (("unmasked", torch.cuda.Stream()), (f"{n} bits, stride {stride}", ops.cu_mask_stream(dev, bits))):
    blocks = 2048
    out = torch.full((blocks, 2), -1, dtype=torch.int32, device=dev)
    with torch.cuda.stream(st):
        _C.check(_C.lib().up_probe_placement(blocks, out.data_ptr(), st.cuda_stream), "probe")
    st.synchronize()
    o = out.cpu().tolist()
    xcc = collections.Counter(r[0] for r in o)
    cus = {(r[0], r[1]) for r in o}
    print(f"{name}: {len(cus)} distinct (XCC, CU/SH/SE) slots; workgroups per XCC: {dict(sorted(xcc.items()))}")
