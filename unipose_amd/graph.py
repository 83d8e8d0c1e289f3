"""Inference forward as ONE hipGraph.

At small batches the forward of the image model is ~130 launches of a few microseconds each and the step is bound by the
host issuing them (B = 1 at 368^2: 3.1 ms wall for 1.3 ms of kernel time).  `GraphedForward` captures one forward —
every convolution (+ folded BatchNorm + ReLU) launch, the pooling / up-sampling / concatenation kernels, the K-split
partial exchange of small launches — into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replays it per call.

    model = unipose(...).cuda().eval()
    fwd = GraphedForward(model, example_input)          # warms up on a private stream, then captures
    heat = fwd(batch)                                   # copies `batch` into the static input, replays, returns the static output

The returned tensors are the graph's static output buffers: consume (or clone) them before the next call.  Shapes are fixed
at capture; the packed weight images the kernels read are captured by address, so call `recapture()` after changing the
parameters (load_state_dict, an optimizer step).  Inference only (`model.eval()`, `torch.no_grad()`).  `close()` (or deleting the object)
releases the graph and — once no other live graph shares the stream handle — the library scratch of its private stream.

Reference call sites: the validation / test loops `unipose.py:150-160`, `uniposeLSTM.py:160-190` (one forward per batch).
"""
import collections
import threading

import torch

# torch hands out stream handles from a 32-entry round-robin pool per device, so two live GraphedForward objects can hold the
# SAME hipStream_t; the library keys its K-split scratch by that handle and a captured graph has the scratch baked in by
# address.  The scratch is therefore released only when the LAST graph captured on a handle closes.
_stream_users = collections.Counter()
_stream_users_lock = threading.Lock()


def _acquire_stream(handle):
    with _stream_users_lock:
        _stream_users[handle] += 1


def _release_stream(handle):
    """True when no other live graph was captured on this handle (the caller may free the library scratch)."""
    with _stream_users_lock:
        _stream_users[handle] -= 1
        if _stream_users[handle] > 0:
            return False
        del _stream_users[handle]
        return True


class GraphedForward:
    def __init__(self, model, *example_args, warmup=3):
        if model.training:
            raise ValueError("GraphedForward captures the inference forward: call model.eval() first")
        tensors = [a for a in example_args if torch.is_tensor(a)]
        if not tensors or not all(t.is_cuda for t in tensors):
            raise TypeError("GraphedForward needs CUDA example inputs (there is no CPU path)")
        self.model = model
        self.device = tensors[0].device
        self.warmup = warmup
        # library-owned scratch (K-split partials, tap-sort tables) is keyed by stream and allocated on first use: warm up
        # on the stream the capture will run on, so nothing allocates while capturing
        self.stream = torch.cuda.Stream(device=self.device)
        _acquire_stream((self.device.index, self.stream.cuda_stream))
        self.static_args = [a.clone() if torch.is_tensor(a) else a for a in example_args]
        self.graph = None
        self.static_out = None
        self.recapture()

    def recapture(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.no_grad(), torch.cuda.stream(self.stream):
            for _ in range(self.warmup):
                self.model(*self.static_args)
        self.stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph, stream=self.stream):
            out = self.model(*self.static_args)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        self.graph, self.static_out = graph, out

    def __call__(self, *args):
        if len(args) != len(self.static_args):
            raise TypeError(f"expected {len(self.static_args)} arguments, got {len(args)}")
        for dst, src in zip(self.static_args, args):
            if torch.is_tensor(dst):
                if not torch.is_tensor(src) or src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError(f"argument of shape {getattr(src, 'shape', None)} / {getattr(src, 'dtype', None)}: the graph was "
                                     f"captured for {tuple(dst.shape)} / {dst.dtype}")
                dst.copy_(src, non_blocking=True)
            elif dst != src:
                raise ValueError(f"non-tensor argument {src!r} differs from the captured {dst!r}")
        self.graph.replay()
        return self.static_out

    def close(self):
        """Drop the graph and hand the private stream's library scratch (K-split partials, 16 MB) back; idempotent."""
        if getattr(self, "stream", None) is None:
            return
        from . import _C
        self.graph, self.static_out = None, None
        torch.cuda.synchronize(self.device)
        if _release_stream((self.device.index, self.stream.cuda_stream)):
            _C.lib().up_stream_release(self.stream.cuda_stream)
        self.stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: the process takes the memory with it
            pass
