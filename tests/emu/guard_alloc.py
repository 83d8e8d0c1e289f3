"""TEST INFRASTRUCTURE: tensors whose last byte sits right in front of an inaccessible page, so that an
out-of-bounds read or write by an emulated kernel crashes the test instead of passing by luck (the GPU
equivalent is a 'Memory access fault' that kills the box)."""
import ctypes
import mmap

import numpy as np
import torch

PAGE = mmap.PAGESIZE
_libc = ctypes.CDLL(None, use_errno=True)
_keep = []


def guarded(shape, dtype=torch.float32, fill=None):
    n = int(np.prod(shape)) if len(shape) else 1
    item = torch.empty((), dtype=dtype).element_size()
    nbytes = max(n * item, 1)
    body = (nbytes + PAGE - 1) // PAGE * PAGE
    m = mmap.mmap(-1, body + PAGE)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
    if _libc.mprotect(ctypes.c_void_p(addr + body), PAGE, 0) != 0:          # PROT_NONE guard page
        raise OSError(ctypes.get_errno(), "mprotect failed")
    start = body - nbytes
    start -= start % 16 if (body - nbytes) % 16 == 0 else 0                  # keep 16-B alignment when sizes allow
    arr = np.frombuffer(m, dtype=np.uint8, count=nbytes, offset=body - nbytes)
    t = torch.from_numpy(arr).view(dtype).reshape(shape)
    _keep.append(m)
    if fill is not None:
        t.fill_(fill)
    return t


class TorchProxy:
    """Drop-in for the `torch` name inside unipose_amd.ops: allocations come from guarded()."""

    def __init__(self):
        self._t = torch

    def __getattr__(self, name):
        return getattr(self._t, name)

    def empty(self, *shape, dtype=torch.float32, device=None, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        return guarded(tuple(shape), dtype, fill=None if dtype != torch.float32 else float("nan"))

    def zeros(self, *shape, dtype=torch.float32, device=None, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        return guarded(tuple(shape), dtype, fill=0)

    def empty_like(self, t, **kw):
        return guarded(tuple(t.shape), t.dtype, fill=None if t.dtype != torch.float32 else float("nan"))

    def zeros_like(self, t, **kw):
        return guarded(tuple(t.shape), t.dtype, fill=0)
